#!/usr/bin/env python3
"""tools/abx.py <lib1.so,lib2.so,...> <workload:blocks,...> [rounds] -- interleaved A/B of several builds of the library:
ONE input and ONE output buffer shared by all builds (no placement effects), the builds take turns launch by launch
(A B C A B C ...: drift of the box hits all alike); per build the minimum, the median and the mean of `rounds` launches.
Differences of 1-2 % on the incompressible workload are inside what two runs of tools/abmulti.py disagree by."""
import ctypes, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
SLOT = 76544
libs = sys.argv[1].split(",")
works = [(w.split(":")[0], int(w.split(":")[1])) for w in sys.argv[2].split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 15
dev = torch.device("cuda:0")
modes = [(p.split("@")[1] if "@" in p else None) for p in libs]  # lib.so@dual: SZ_KERNEL=dual for that build's launches; lib.so@noadopt: SZ_NO_ADOPT=1
libs = [p.split("@")[0] for p in libs]
L = []
for p in libs:
    l = ctypes.CDLL(os.path.join(ROOT, p) if not os.path.isabs(p) else p)
    vp, sz, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
    l.sz_compress_blocks_hip.argtypes = [vp, sz, vp, u32, vp, sz, vp, u32, vp]
    l.sz_compress_blocks_hip.restype = ctypes.c_int
    L.append(l)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for kind, nb in works:
    src = bench.make_input(kind, 0, nb, dev)
    dst = torch.zeros(nb * SLOT, dtype=torch.uint8, device=dev)
    ol = torch.zeros(nb, dtype=torch.int32, device=dev)
    def run(l):
        m = modes[L.index(l)]
        os.environ.pop("SZ_KERNEL", None)
        os.environ.pop("SZ_NO_ADOPT", None)
        if m == "noadopt":
            os.environ["SZ_NO_ADOPT"] = "1"
        elif m:
            os.environ["SZ_KERNEL"] = m
        rc = l.sz_compress_blocks_hip(src.data_ptr(), 65536, None, nb, dst.data_ptr(), SLOT, ol.data_ptr(), 1, stream)
        assert rc == 0, rc
    for l in L:
        for _ in range(2):
            run(l)
    torch.cuda.synchronize()
    ts = [[] for _ in L]
    for r in range(rounds):
        for i, l in enumerate(L):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(l); b.record(); torch.cuda.synchronize()
            ts[i].append(a.elapsed_time(b))
    for p, m, t in zip(libs, modes, ts):
        t.sort()
        p = p + ("@" + m if m else "")
        print("%-30s %-5s nb=%-6d min %.3f  med %.3f  mean %.3f ms -> %.1f GB/s (median)" % (
            os.path.basename(p), kind, nb, t[0], t[len(t) // 2], statistics.mean(t), nb * 65536 / t[len(t) // 2] / 1e6), flush=True)
    del src, dst, ol
