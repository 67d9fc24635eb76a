#!/usr/bin/env python3
"""tools/graphbench.py -- direct launches vs hipGraph replay of one compress step (C3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from pkg_loader import load_package
P = load_package()
kind = sys.argv[1] if len(sys.argv) > 1 else "c3"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
dev = torch.device("cuda:0")
src = bench.make_input(kind, 0, nb, dev)
dst = torch.empty(nb * P.SLOT_STRIDE, dtype=torch.uint8, device=dev)
ol = torch.zeros(nb, dtype=torch.int32, device=dev)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol, stream=s)
s.synchronize()
def wall(f, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.cuda.stream(s):
    d = wall(lambda: P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol, stream=s))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol, stream=s)
r = wall(lambda: g.replay())
print("%s nb=%d: direct %.4f ms/step, graph replay %.4f ms/step" % (kind, nb, d, r))
