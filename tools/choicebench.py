#!/usr/bin/env python3
"""tools/choicebench.py -- does the launch sample choose the faster kernel on data between the bench's extremes?
For several kinds of synthetic 64 KiB blocks: ms of the default (sampled) choice, of SZ_KERNEL=single and of
SZ_KERNEL=team on the same launch, and the compression ratio."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from pkg_loader import load_package
P = load_package()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(7)
def alphabet(k):
    return (torch.randint(0, k, (nb * 65536,), device=dev, generator=g, dtype=torch.int32) + 48).to(torch.uint8)
def sprinkled(frac, ln):
    """random bytes; a fraction of the positions overwritten by copies of a string seen `ln`..64 KiB earlier"""
    a = torch.randint(0, 256, (nb, 65536), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
    n_cp = int(65536 * frac / ln)
    for _ in range(n_cp):
        s = int(torch.randint(0, 65536 - 2 * ln, (1,)).item()); d = int(torch.randint(s + ln, 65536 - ln, (1,)).item())
        a[:, d:d + ln] = a[:, s:s + ln]
    return a.reshape(-1)
kinds = [("alphabet 4", lambda: alphabet(4)), ("alphabet 16", lambda: alphabet(16)), ("alphabet 64", lambda: alphabet(64)),
         ("random + 2 % copies of 32 B", lambda: sprinkled(0.02, 32)), ("random + 10 % copies of 32 B", lambda: sprinkled(0.10, 32)),
         ("random + 30 % copies of 64 B", lambda: sprinkled(0.30, 64)), ("random + 10 % copies of 512 B", lambda: sprinkled(0.10, 512))]
for name, make in kinds:
    src = make()
    dst = torch.empty(nb * P.SLOT_STRIDE, dtype=torch.uint8, device=dev); ol = torch.zeros(nb, dtype=torch.int32, device=dev)
    res = {}
    for mode in ("auto", "single", "team"):
        if mode == "auto": os.environ.pop("SZ_KERNEL", None)
        else: os.environ["SZ_KERNEL"] = mode
        for _ in range(2): P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol)
        torch.cuda.synchronize(); ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res[mode] = min(ts)
    ratio = nb * 65536 / float(ol.sum().item())
    best = min(res["single"], res["team"])
    print("%-32s ratio %.3f  auto %.3f ms  single %.3f  team %.3f  -> auto is %.2fx the better of the two" % (name, ratio, res["auto"], res["single"], res["team"], res["auto"] / best), flush=True)
    del src, dst, ol
