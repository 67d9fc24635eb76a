import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from pkg_loader import load_package
P = load_package()
nb=32768; dev=torch.device("cuda:0")
src = bench.make_input("c3", 0, nb, dev)
dst = torch.empty(nb * P.SLOT_STRIDE, dtype=torch.uint8, device=dev)
ol = torch.zeros(nb, dtype=torch.int32, device=dev)
steps=40
hev = bench.HipEvents(steps)
def run(with_ev, n_ev):
    for _ in range(3): P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(steps):
        if with_ev:
            e=hev.ev[i]
            if n_ev==4: P.profile_next_launch(*e)
            else: P.profile_next_launch(e[0], e[1], None, None)
        P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol)
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/steps*1e3
for r in range(3):
    print("no events %.4f ms   2 events %.4f ms   4 events %.4f ms" % (run(False,0), run(True,2), run(True,4)), flush=True)
