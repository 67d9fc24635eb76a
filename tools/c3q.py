import ctypes, os, sys, statistics
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
SLOT=76544
lib = ctypes.CDLL(os.path.join(ROOT, "compression-accelerator_amd", "csrc", "libsnappy_hip.so"))
vp, sz, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
lib.sz_compress_blocks_hip.argtypes = [vp, sz, vp, u32, vp, sz, vp, u32, vp]
lib.sz_compress_blocks_hip.restype = ctypes.c_int
dev=torch.device("cuda:0"); stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for nb in (32768, 16384, 65536):
    src = bench.make_input("c3", 0, nb, dev)
    dst = torch.zeros(nb * SLOT, dtype=torch.uint8, device=dev); ol = torch.zeros(nb, dtype=torch.int32, device=dev)
    ref=None
    def run(mode):
        os.environ["SZ_C3_KERNEL"]=mode
        rc = lib.sz_compress_blocks_hip(src.data_ptr(), 65536, None, nb, dst.data_ptr(), SLOT, ol.data_ptr(), 1, stream); assert rc==0
    for m in ("block","queue"): run(m); run(m)
    torch.cuda.synchronize()
    run("block"); torch.cuda.synchronize(); ref=(dst.clone(), ol.clone())
    dst.zero_(); ol.zero_(); run("queue"); torch.cuda.synchronize()
    idx = torch.arange(SLOT, device=dev)[None, :] < ref[1][:, None]
    same = bool(torch.equal(ol, ref[1])) and bool(torch.equal(torch.where(idx, dst.view(nb, SLOT), 0), torch.where(idx, ref[0].view(nb, SLOT), 0)))
    ts={"block":[], "queue":[]}
    for r in range(21):
        for m in ("block","queue"):
            a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            a.record(); run(m); b.record(); torch.cuda.synchronize(); ts[m].append(a.elapsed_time(b))
    for m in ts:
        t=sorted(ts[m]); print("nb=%6d %-6s min %.3f med %.3f mean %.3f ms -> %.1f GB/s  same=%s"%(nb,m,t[0],t[len(t)//2],statistics.mean(t),nb*65536/t[len(t)//2]/1e6,same),flush=True)
    del src,dst,ol,ref
