#!/usr/bin/env python3
"""tools/soak.py -- long differential run of the GPU compressor against the oracle (GPU box).
usage: soak.py [n_cases] [seed]   -- every case through the default choice (batches of 2 500 mixed blocks: sampled, and routed
block by block since round 5), the single, the team and the multi-wave kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzzgen, oracle_lib
from gpu_util import compress_many
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 123456
rs = np.random.RandomState(seed0)
edges = [0, 1, 14, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 65, 79, 80, 81, 95, 96, 97, 255, 256, 257, 4095, 4096, 4097,
         16383, 16384, 16385, 32767, 32768, 65519, 65520, 65521, 65535, 65536]
bad = 0; done = 0; t0 = time.time()
BATCH = 2500
while done < n_cases:
    cases = []
    for i in range(min(BATCH, n_cases - done)):
        kind = fuzzgen.KINDS[int(rs.randint(0, len(fuzzgen.KINDS)))]
        r = rs.randint(0, 10)
        n = edges[int(rs.randint(0, len(edges)))] if r < 3 else int(rs.randint(0, 2000)) if r < 6 else int(rs.randint(0, 65537))
        cases.append((kind, seed0 + done + i, n))
    blocks = [fuzzgen.gen(k, s, n).tobytes() for k, s, n in cases]
    want = [oracle_lib.raw_compress(b) for b in blocks]
    for mode in ("auto", "single", "team", "dual"):
        if mode == "auto":
            os.environ.pop("SZ_KERNEL", None)
        else:
            os.environ["SZ_KERNEL"] = mode
        got = compress_many(blocks)
        for c, g, w in zip(cases, got, want):
            if g != w:
                bad += 1
                print("MISMATCH", mode, c, len(g), len(w), flush=True)
    # GPU decompression of the oracle's streams must give the inputs back
    import torch
    from pkg_loader import load_package
    P = load_package()
    nbk = len(blocks)
    host = np.zeros((nbk, P.SLOT_STRIDE), dtype=np.uint8); clen = np.zeros(nbk, dtype=np.int32)
    for i, c in enumerate(want):
        host[i, :len(c)] = np.frombuffer(c, dtype=np.uint8); clen[i] = len(c)
    d_src = torch.from_numpy(host).cuda().reshape(-1); d_len = torch.from_numpy(clen).cuda()
    dstb, ulen, status = P.uncompress_blocks(d_src, P.SLOT_STRIDE, d_len, nbk)
    torch.cuda.synchronize()
    ul = ulen.cpu().numpy(); stt = status.cpu().numpy(); outb = dstb.cpu().numpy().reshape(nbk, 65536)
    for i, b in enumerate(blocks):
        if stt[i] != 0 or ul[i] != len(b) or outb[i, :len(b)].tobytes() != b:
            bad += 1
            print("DECOMPRESS MISMATCH", cases[i], int(stt[i]), int(ul[i]), flush=True)
    done += len(cases)
    print("%d cases, %d mismatches, %.0f s" % (done, bad, time.time() - t0), flush=True)
print("soak done: %d cases x 4 modes, %d mismatches" % (done, bad))
sys.exit(1 if bad else 0)
