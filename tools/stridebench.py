#!/usr/bin/env python3
"""tools/stridebench.py [blocks] [rounds] -- the incompressible workload at several output slot strides, interleaved
(one input, one output buffer sized for the largest stride): does the distance between the slots matter to the HBM?"""
import ctypes, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 15
strides = [76544, 76800, 77824, 78848, 81920, 86016, 98304, 131072]
lib = ctypes.CDLL(os.path.join(ROOT, "compression-accelerator_amd", "csrc", "libsnappy_hip.so"))
vp, sz, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
lib.sz_compress_blocks_hip.argtypes = [vp, sz, vp, u32, vp, sz, vp, u32, vp]
lib.sz_compress_blocks_hip.restype = ctypes.c_int
dev = torch.device("cuda:0")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
src = bench.make_input("c3", 0, nb, dev)
dst = torch.zeros(nb * max(strides), dtype=torch.uint8, device=dev)
ol = torch.zeros(nb, dtype=torch.int32, device=dev)
def run(st):
    rc = lib.sz_compress_blocks_hip(src.data_ptr(), 65536, None, nb, dst.data_ptr(), st, ol.data_ptr(), 1, stream)
    assert rc == 0, rc
for st in strides:
    run(st); run(st)
torch.cuda.synchronize()
ts = {st: [] for st in strides}
for r in range(rounds):
    for st in strides:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(st); b.record(); torch.cuda.synchronize()
        ts[st].append(a.elapsed_time(b))
for st in strides:
    t = sorted(ts[st])
    print("stride %6d (%6.2f KiB)  min %.3f  med %.3f  mean %.3f ms -> %.1f GB/s input (median)" % (
        st, st / 1024, t[0], t[len(t) // 2], statistics.mean(t), nb * 65536 / t[len(t) // 2] / 1e6), flush=True)
