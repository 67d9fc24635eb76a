#!/usr/bin/env python3
"""tools/c3ab.py [blocks] [reps] [kind] -- the two kernels for launches that the sample finds incompressible
(SZ_C3_KERNEL=single|stream) in one process on the same input: kernel-only time, and every block of the second
against the first (lengths and the bytes below them)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
SLOT = 76544
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
kind = sys.argv[3] if len(sys.argv) > 3 else "c3"
lib = ctypes.CDLL(os.path.join(ROOT, "compression-accelerator_amd", "csrc", "libsnappy_hip.so"))
vp, sz, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
lib.sz_compress_blocks_hip.argtypes = [vp, sz, vp, u32, vp, sz, vp, u32, vp]
lib.sz_compress_blocks_hip.restype = ctypes.c_int
dev = torch.device("cuda:0")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
src = bench.make_input(kind, 0, nb, dev)
ref = None
for rnd in range(2):
    for mode in ("single", "stream"):
        os.environ["SZ_C3_KERNEL"] = mode
        dst = torch.full((nb * SLOT,), 0xA5, dtype=torch.uint8, device=dev)
        ol = torch.zeros(nb, dtype=torch.int32, device=dev)
        def run():
            rc = lib.sz_compress_blocks_hip(src.data_ptr(), 65536, None, nb, dst.data_ptr(), SLOT, ol.data_ptr(), 1, stream)
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
        if ref is None:
            ref = (dst.clone(), ol.clone())
        else:
            same_len = bool(torch.equal(ol, ref[1]))
            idx = torch.arange(SLOT, device=dev)[None, :] < ref[1][:, None]
            same = same_len and bool(torch.equal(torch.where(idx, dst.view(nb, SLOT), 0), torch.where(idx, ref[0].view(nb, SLOT), 0)))
            msg = "  same_as_first=%s" % same
        print("%-7s %-5s nb=%-6d min %.3f ms  med %.3f ms  -> %.1f GB/s input%s" % (mode, kind, nb, ts[0], ts[len(ts) // 2], nb * 65536 / ts[0] / 1e6, msg), flush=True)
        del dst, ol
