#!/usr/bin/env python3
"""tools/mixcost.py [blocks] -- the performance bar that used to sit in the GPU suite (tests/test_gpu_routing.py keeps the
correctness half): a launch of `blocks` blocks with every 16th block text costs <= 1.25 x (its incompressible part alone
+ its text part alone).  VERDICT r4 item 3; round 5: 1.08 x.  Exit status 1 above the bar."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
import bench
from pkg_loader import load_package
P = load_package()
dev = torch.device("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
c3 = bench.make_input("c3", 0, nb, dev).reshape(nb, 65536)
text = bench.make_input("text", 5, nb // 16, dev).reshape(-1, 65536)
mixed = c3.clone()
mixed[0::16] = text
keep = torch.ones(nb, dtype=torch.bool, device=dev)
keep[0::16] = False


def time_of(src, n):
    src = src.reshape(-1).contiguous()
    dst = torch.empty(n * P.SLOT_STRIDE, dtype=torch.uint8, device=dev)
    ol = torch.zeros(n, dtype=torch.int32, device=dev)
    ts = []
    for i in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); P.compress_blocks(src, 65536, n, dst=dst, out_len=ol); b.record(); torch.cuda.synchronize()
        if i >= 2:
            ts.append(a.elapsed_time(b))
    return min(ts)


t_mixed, t_c3, t_text = time_of(mixed, nb), time_of(c3[keep], nb - nb // 16), time_of(text, nb // 16)
ratio = t_mixed / (t_c3 + t_text)
print("mixed %.3f ms, incompressible part %.3f ms, text part %.3f ms: ratio %.3f (bar 1.25)" % (t_mixed, t_c3, t_text, ratio))
sys.exit(0 if ratio <= 1.25 else 1)
