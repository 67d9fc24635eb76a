#!/usr/bin/env python3
"""tools/hostrate.py -- PCIe-inclusive rate of the host-buffer API (sz_compress through the snappy:: facade
driver): builds drivers/sw_benchmark.cc and runs it on a tiled corpus file.  usage: hostrate.py [MiB]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import build_util, fuzzgen
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = tempfile.mkdtemp()
exe = build_util.cxx(os.path.join(build_util.DRIVERS, "sw_benchmark.cc"), os.path.join(d, "sw_benchmark"))
for name, corpus in (("text", fuzzgen.corpus("mtg")), ("random", fuzzgen.corpus("random_ascii"))):
    path = os.path.join(d, name + ".bin")
    with open(path, "wb") as f:
        total = mib << 20
        reps = total // len(corpus) + 1
        f.write((corpus * reps)[:total])
    out = subprocess.run([exe, name, path], capture_output=True, text=True, timeout=900).stdout.split()
    t, length, ns, clen = out[1].split(";")
    print("%s: %d MiB -> %s bytes, %.1f ms per snappy::Compress call = %.2f GB/s (H2D + kernels + pack + D2H, pageable host memory)"
          % (name, mib, clen, int(ns) / 1e6, int(length) / int(ns)))
