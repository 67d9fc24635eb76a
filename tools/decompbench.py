#!/usr/bin/env python3
"""tools/decompbench.py -- throughput of the batch decompressor (GPU box).  usage: decompbench.py <kind> [n_blocks]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from pkg_loader import load_package
P = load_package()
kind = sys.argv[1]; nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda:0")
src = bench.make_input(kind, 0, nb, dev)
dst, ol = P.compress_blocks(src, 65536, nb)
torch.cuda.synchronize()
back = torch.empty(nb * 65536, dtype=torch.uint8, device=dev)
ts = []
for _ in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); P.uncompress_blocks(dst, P.SLOT_STRIDE, ol, nb, dst=back); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ms = min(ts)
ok = torch.equal(back, src)
print("decompress %s nb=%d: best %.3f ms -> %.2f GB/s (uncompressed bytes), round trip %s" % (kind, nb, ms, nb * 65536 / ms / 1e6, ok))
