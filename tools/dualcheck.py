#!/usr/bin/env python3
"""tools/dualcheck.py -- small, chatty check of the forced multi-wave kernel (run under `timeout`)."""
import os, sys
os.environ["SZ_KERNEL"] = "dual"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fuzzgen, oracle_lib
from gpu_util import compress_many
def say(*a):
    print(*a, flush=True)
cases = [("random", 100), ("text_mtg", 100), ("text_mtg", 4096), ("text_mtg", 65536), ("random", 65536), ("lz", 65536)]
for kind, n in cases:
    blocks = [fuzzgen.gen(kind, 5 + i, n).tobytes() for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2)]
    say("case", kind, n, "launch...")
    from pkg_loader import load_package
    P = load_package()
    nb = len(blocks)
    host = np.zeros((nb, 65536), dtype=np.uint8); lens = np.zeros(nb, dtype=np.int32)
    for i, b in enumerate(blocks):
        host[i, :len(b)] = np.frombuffer(b, dtype=np.uint8); lens[i] = len(b)
    src = torch.from_numpy(host).cuda().reshape(-1); il = torch.from_numpy(lens).cuda()
    dst, ol = P.compress_blocks(src, 65536, nb, in_len=il)
    torch.cuda.synchronize()
    olh = ol.cpu().numpy().astype(np.uint32)
    out = dst.cpu().numpy().reshape(nb, P.SLOT_STRIDE)
    ok = [(int(olh[i]) < 80000 and out[i, :olh[i]].tobytes() == oracle_lib.raw_compress(blocks[i])) for i in range(nb)]
    say("   out_len", [hex(int(x)) for x in olh], "ok", ok)
say("done")
