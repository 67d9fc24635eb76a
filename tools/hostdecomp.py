#!/usr/bin/env python3
"""tools/hostdecomp.py -- rate of the host-buffer decompression (sz_uncompress = snappy::Uncompress) on ONE
raw stream of the given size: the stream is a single wave's work.  usage: hostdecomp.py [MiB]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzzgen
from pkg_loader import load_package
P = load_package()
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for name, corpus in (("text", fuzzgen.corpus("mtg")), ("random", fuzzgen.corpus("random_ascii"))):
    total = mib << 20
    data = (corpus * (total // len(corpus) + 1))[:total]
    comp = P.compress(data)
    P.uncompress(comp)
    t0 = time.perf_counter(); back = P.uncompress(comp); dt = time.perf_counter() - t0
    print("%s: one stream of %d MiB (%d compressed bytes): sz_uncompress %.1f ms = %.3f GB/s of output, %s"
          % (name, mib, len(comp), dt * 1e3, total / dt / 1e9, "ok" if back == data else "MISMATCH"), flush=True)
