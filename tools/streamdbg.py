#!/usr/bin/env python3
"""tools/streamdbg.py <lib built with -DSZ_STREAM_DEBUG> [MiB] -- runs a 1-stream decode through a debug build; the
kernels print the serial pass's counters and the most 64-byte windows any wave of the boundaries pass walked (the
pass is as slow as its slowest wave) with that wave's segment.  The API's status words are never touched by it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fuzzgen
from pkg_loader import load_package
P = load_package()
P.LIB_PATH = os.path.join(ROOT, sys.argv[1])
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
for name, corpus in (("text", fuzzgen.corpus("mtg")), ("random", fuzzgen.corpus("random_ascii"))):
    total = mib << 20
    data = (corpus * (total // len(corpus) + 1))[:total]
    comp = P.compress(data)
    n = len(comp)
    body = 1
    while comp[body - 1] & 0x80:
        body += 1
    buf = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    buf[:n] = torch.from_numpy(np.frombuffer(comp, dtype=np.uint8).copy()).cuda()
    dst = torch.empty(total, dtype=torch.uint8, device="cuda")
    _, st = P.uncompress_stream(buf, n, body, total, dst=dst)
    st = st.cpu().numpy().tolist()
    torch.cuda.synchronize()
    print("%s: status words %s" % (name, st), flush=True)
