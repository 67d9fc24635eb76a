#!/usr/bin/env python3
"""tools/decompab.py <lib1.so,lib2.so,...> <kind:blocks,...> -- batch decompression rate of several builds of the library
in one process (compressed once by the first build), round trip checked."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from pkg_loader import load_package
P = load_package()
libs = sys.argv[1].split(",")
works = [(w.split(":")[0], int(w.split(":")[1])) for w in sys.argv[2].split(",")]
dev = torch.device("cuda:0")
vp, sz, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
L = []
for p in libs:
    l = ctypes.CDLL(os.path.join(ROOT, p))
    l.sz_uncompress_blocks_hip.argtypes = [vp, sz, vp, u32, vp, sz, vp, vp, vp]
    l.sz_uncompress_blocks_hip.restype = ctypes.c_int
    L.append(l)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for kind, nb in works:
    src = bench.make_input(kind, 0, nb, dev)
    dst, ol = P.compress_blocks(src, 65536, nb)
    torch.cuda.synchronize()
    for p, l in zip(libs, L):
        back = torch.zeros(nb * 65536, dtype=torch.uint8, device=dev)
        ulen = torch.zeros(nb, dtype=torch.int32, device=dev)
        st = torch.zeros(nb, dtype=torch.int32, device=dev)
        def run():
            rc = l.sz_uncompress_blocks_hip(dst.data_ptr(), P.SLOT_STRIDE, ol.data_ptr(), nb, back.data_ptr(), 65536, ulen.data_ptr(), st.data_ptr(), stream)
            assert rc == 0, rc
        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ok = bool(torch.equal(back, src)) and int(st.abs().sum().item()) == 0
        print("%-28s %-5s nb=%d  min %.3f ms -> %.1f GB/s  round trip %s" % (os.path.basename(p), kind, nb, min(ts), nb * 65536 / min(ts) / 1e6, ok), flush=True)
