import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, bench
from pkg_loader import load_package
P = load_package()
P.LIB_PATH = sys.argv[1]
kind, nb = sys.argv[2], int(sys.argv[3])
dev = torch.device("cuda:0")
src = bench.make_input(kind, 0, nb, dev)
dst = torch.empty(nb * P.SLOT_STRIDE, dtype=torch.uint8, device=dev); ol = torch.zeros(nb, dtype=torch.int32, device=dev)
for _ in range(3): P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol)
torch.cuda.synchronize(); ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort(); print(os.path.basename(sys.argv[1]), kind, "min %.4f med %.4f ms" % (ts[0], ts[len(ts)//2]), flush=True)
