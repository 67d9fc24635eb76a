#!/usr/bin/env python3
"""tools/quickbench.py -- kernel-only timing of one workload (for A/B experiments on the GPU box).
usage: python tools/quickbench.py <c3|text|c5|mix> [n_blocks] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from pkg_loader import load_package
P = load_package()
if os.environ.get("SZ_AB_LIB"):  # A/B experiments: another build of the library
    P.LIB_PATH = os.path.join(ROOT, os.environ["SZ_AB_LIB"])
kind = sys.argv[1]; nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4096; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
src = bench.make_input(kind, 0, nb, dev)
dst = torch.empty(nb * P.SLOT_STRIDE, dtype=torch.uint8, device=dev)
ol = torch.zeros(nb, dtype=torch.int32, device=dev)
for _ in range(2):
    P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ms = min(ts)
print("%s nb=%d flags=%s: best %.3f ms avg %.3f ms -> %.2f GB/s input" % (kind, nb, os.environ.get("SZ_DEBUG_FLAGS", "0"), ms, sum(ts)/len(ts), nb * 65536 / ms / 1e6))
