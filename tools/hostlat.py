#!/usr/bin/env python3
"""tools/hostlat.py -- latency of the host-buffer API on small inputs (the RoCC surface compresses one buffer per
command: tests/compression.c does 130 bytes): microseconds per sz_compress / sz_uncompress call, median of 200."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzzgen
from pkg_loader import load_package
P = load_package()
text = fuzzgen.corpus("mtg")
for n in (130, 4602, 65536, 200000, 1 << 20):
    data = bytes(text[:n])
    comp = P.compress(data)
    for _ in range(5):
        P.compress(data); P.uncompress(comp)
    tc, tu = [], []
    for _ in range(200 if n <= 65536 else 50):
        t0 = time.perf_counter(); P.compress(data); t1 = time.perf_counter(); P.uncompress(comp); t2 = time.perf_counter()
        tc.append(t1 - t0); tu.append(t2 - t1)
    tc.sort(); tu.sort()
    print("n=%8d  compress %8.1f us  uncompress %8.1f us  (%d -> %d bytes)" % (n, tc[len(tc) // 2] * 1e6, tu[len(tu) // 2] * 1e6, n, len(comp)), flush=True)
