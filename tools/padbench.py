#!/usr/bin/env python3
"""tools/padbench.py <lib built with -DSZ_ABLATION> [workload:blocks] [rounds] -- what is one instruction worth?
The build pads a tile with 64 independent vector (0x1000 / 0x2000 / 0x4000) or scalar (0x10000 / 0x20000 / 0x40000)
instructions in the lone waves' tile loop / off the token's path of a team's turn / on it (csrc/diag.h: ABLATION_PAD);
output bytes unchanged.  Interleaved launches on one pair of buffers; prints the rate per setting and the slowdown per
64 instructions (EXPERIMENTS.md R5.3)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
SLOT = 76544
lib = ctypes.CDLL(os.path.join(ROOT, sys.argv[1]))
kind, nb = (sys.argv[2].split(":")[0], int(sys.argv[2].split(":")[1])) if len(sys.argv) > 2 else ("text", 16384)
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
vp, sz, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
lib.sz_compress_blocks_hip.argtypes = [vp, sz, vp, u32, vp, sz, vp, u32, vp]
dev = torch.device("cuda:0")
src = bench.make_input(kind, 0, nb, dev)
dst = torch.zeros(nb * SLOT, dtype=torch.uint8, device=dev)
ol = torch.zeros(nb, dtype=torch.int32, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
settings = [("none", 0), ("lone loop, 64 vector", 0x1000), ("team off-path, 64 vector", 0x2000), ("team on-path, 64 vector", 0x4000),
            ("lone loop, 64 scalar", 0x10000), ("team off-path, 64 scalar", 0x20000), ("team on-path, 64 scalar", 0x40000),
            ("all three, 64 vector each", 0x7000)]
def run(f):
    os.environ["SZ_DEBUG_FLAGS"] = hex(f)
    assert lib.sz_compress_blocks_hip(src.data_ptr(), 65536, None, nb, dst.data_ptr(), SLOT, ol.data_ptr(), 1, stream) == 0
ref = None
ts = {n: [] for n, _ in settings}
for n, f in settings:
    run(f); torch.cuda.synchronize()
    if ref is None:
        ref = ol.clone()
    assert torch.equal(ol, ref), "padding changed the output lengths: " + n
for r in range(rounds):
    for n, f in settings:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(f); b.record(); torch.cuda.synchronize()
        ts[n].append(a.elapsed_time(b))
base = sorted(ts["none"])[rounds // 2]
for n, f in settings:
    t = sorted(ts[n])[rounds // 2]
    print("%-28s %s nb=%d  med %.3f ms -> %.1f GB/s  (%+.1f %% time)" % (n, kind, nb, t, nb * 65536 / t / 1e6, 100 * (t / base - 1)), flush=True)
