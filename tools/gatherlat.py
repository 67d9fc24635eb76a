#!/usr/bin/env python3
"""tools/gatherlat.py [blocks_per_rank] -- what the path's one collective costs where it can be measured on this pool: ONE
rank.  `all_gather_into_tensor` of `blocks_per_rank` uint32 lengths through RCCL (backend nccl, world size 1: the
collective is launched, there is no peer), and the same followed by the prefix sum (sharding.gather_lengths' work):
HIP-event time per call on the stream it runs on, and the host's time to enqueue it.  DESIGN.md section 5 builds its
expectation for 2, 4 and 8 GPUs on these two numbers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
lens = torch.randint(1000, 70000, (n,), dtype=torch.int32, device=dev)
out = torch.empty(n, dtype=torch.int32, device=dev)
offs = torch.empty(n + 1, dtype=torch.int64, device=dev)


def gather_only():
    dist.all_gather_into_tensor(out, lens)


def gather_and_scan():
    dist.all_gather_into_tensor(out, lens)
    offs[:1].zero_()
    torch.cumsum(out, dim=0, dtype=torch.int64, out=offs[1:])


for name, fn in (("all_gather_into_tensor (RCCL, 1 rank)", gather_only), ("all_gather + prefix sum", gather_and_scan)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ev, host = [], []
    for _ in range(200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); a.record(); fn(); b.record(); host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        ev.append(a.elapsed_time(b))
    ev.sort(); host.sort()
    print("%-40s %d uint32: device median %.1f us (min %.1f), host enqueue median %.1f us" % (name, n, ev[100] * 1e3, ev[0] * 1e3, host[100] * 1e6), flush=True)
dist.destroy_process_group()
