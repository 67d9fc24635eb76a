#!/usr/bin/env python3
"""tools/abmulti.py -- kernel-only timing of several builds of the library in one process (A/B on the GPU box).
usage: python tools/abmulti.py <lib1.so,lib2.so,...> <workload:blocks,...> [reps] [--check]
--check compares every block of every build with the first build's output (bytes and lengths)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
SLOT = 76544
libs = sys.argv[1].split(",")
works = [(w.split(":")[0], int(w.split(":")[1])) for w in sys.argv[2].split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 5
check = "--check" in sys.argv
dev = torch.device("cuda:0")
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
    ref = None
    for p, l in zip(libs, L):
        dst = torch.zeros(nb * SLOT, dtype=torch.uint8, device=dev)
        ol = torch.zeros(nb, dtype=torch.int32, device=dev)
        def run():
            rc = l.sz_compress_blocks_hip(src.data_ptr(), 65536, None, nb, dst.data_ptr(), SLOT, ol.data_ptr(), 1, stream)
            assert rc == 0, rc
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        msg = ""
        if check:
            if ref is None:
                ref = (dst.clone(), ol.clone())
            else:
                same_len = bool(torch.equal(ol, ref[1]))
                idx = torch.arange(SLOT, device=dev)[None, :] < ref[1][:, None]
                same = same_len and bool(torch.equal(torch.where(idx, dst.view(nb, SLOT), 0), torch.where(idx, ref[0].view(nb, SLOT), 0)))
                msg = "  same_as_first=%s" % same
        print("%-28s %-5s nb=%-6d min %.3f ms  med %.3f ms  -> %.1f GB/s%s" % (os.path.basename(p), kind, nb, ts[0], ts[len(ts) // 2], nb * 65536 / ts[0] / 1e6, msg), flush=True)
        del dst, ol
    del src
