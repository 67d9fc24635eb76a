import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
import bench, oracle_lib
from pkg_loader import load_package
P = load_package()
dev = torch.device("cuda:0")
nb = 262144
t0 = time.time()
src = bench.make_input("c3", 0, nb, dev)   # 16 GiB
torch.cuda.synchronize(); print("input ready %.1fs, free %.1f GiB" % (time.time() - t0, torch.cuda.mem_get_info()[0] / 2**30), flush=True)
dst = torch.empty(nb * P.SLOT_STRIDE, dtype=torch.uint8, device=dev)
ol = torch.zeros(nb, dtype=torch.int32, device=dev)
for _ in range(2):
    P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol); b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b)
print("16 GiB C3 on one GPU: %.2f ms -> %.0f GB/s; lens min %d max %d" % (ms, nb * 65536 / ms / 1e6, int(ol.min()), int(ol.max())), flush=True)
back, ul, st = P.uncompress_blocks(dst, P.SLOT_STRIDE, ol, nb)
torch.cuda.synchronize()
print("round trip:", bool(torch.equal(back, src)), "status sum", int(st.abs().sum()))
