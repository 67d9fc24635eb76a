#!/usr/bin/env python3
"""tools/hwrate.py [kind] [blocks] -- rate and ratio of SZ_FLAG_HW_COMPAT (the reference accelerator's own element
stream, one wave per block, serial parse) beside the default (software Snappy's bytes) on the same blocks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from pkg_loader import load_package
P = load_package()
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
src = bench.make_input(kind, 0, nb, torch.device("cuda:0"))
for hw in (False, True):
    dst, ol = P.compress_blocks(src, 65536, nb, hw_compat=hw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol, hw_compat=hw); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    back, ulen, st = P.uncompress_blocks(dst, P.SLOT_STRIDE, ol, nb)
    ok = bool(torch.equal(back, src)) and int(st.abs().sum().item()) == 0
    print("%s nb=%d %s: %.3f ms = %.2f GB/s, ratio %.3f, decodes back: %s" % (kind, nb, "accelerator's format" if hw else "Snappy 1.1.8 bytes   ",
          min(ts), nb * 65536 / min(ts) / 1e6, nb * 65536 / float(ol.sum().item()), ok), flush=True)
