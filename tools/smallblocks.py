#!/usr/bin/env python3
"""tools/smallblocks.py -- throughput on blocks smaller than 64 KiB (e.g. 4 KiB pages).  usage: smallblocks.py <kind> <block_bytes> [MiB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from pkg_loader import load_package
P = load_package()
kind, bs = sys.argv[1], int(sys.argv[2]); mib = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
total = mib << 20
src = bench.make_input(kind, 0, total // 65536, dev)
nb = total // bs
stride = int(P.max_compressed_length(bs)) + 8
stride = (stride + 15) // 16 * 16
dst = torch.empty(nb * stride, dtype=torch.uint8, device=dev); ol = torch.zeros(nb, dtype=torch.int32, device=dev)
for _ in range(2): P.compress_blocks(src, bs, nb, dst=dst, dst_stride=stride, out_len=ol)
torch.cuda.synchronize(); ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); P.compress_blocks(src, bs, nb, dst=dst, dst_stride=stride, out_len=ol); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ms = min(ts)
print("%s blocks of %d B x %d: %.3f ms -> %.2f GB/s, ratio %.3f" % (kind, bs, nb, ms, total / ms / 1e6, total / float(ol.sum().item())))
