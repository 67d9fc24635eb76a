#!/usr/bin/env python3
"""bench.py -- Snappy block compression throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|text|c5|mix] [--blocks-per-gpu B]

A "step" is one pass of the hot path (sz_compress_blocks_hip through the C ABI) over this rank's
blocks, input already resident in HBM, plus -- for N>1 -- the RCCL all_gather of the per-block
lengths.  Prints ONE JSON line on rank 0.  N>1 is launched by torch.distributed.run (one process
per GPU); blocks shard by contiguous ranges, weak scaling (fixed blocks per GPU).

Headline workload (config.workload): C3 = consecutive 64 KiB windows of the infinite tiling of
data/randomASCII.txt (BASELINE.json configs[2], the one the north-star target is quoted on);
--blocks-per-gpu 32768 (2 GiB per GPU) so that N=8 is config C4's 16 GiB.  Short passes over the
text (C2-like), low-entropy (C5) and mixed (C4 mix) workloads are reported in "workloads".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

BLOCK = 65536
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def make_input(kind, first_block, n_blocks, device):
    """Builds this rank's blocks [first_block, first_block+n_blocks) directly in HBM."""
    import torch
    import fuzzgen

    def tiling(src_bytes, start, total):
        src = torch.from_numpy(np.frombuffer(src_bytes, dtype=np.uint8).copy()).to(device)
        L = src.numel()
        start %= L
        reps = (start + total + L - 1) // L
        return src.repeat(reps)[start:start + total].contiguous()

    total = n_blocks * BLOCK
    if kind == "c3":
        return tiling(fuzzgen.corpus("random_ascii"), first_block * BLOCK, total)
    if kind == "text":
        return tiling(fuzzgen.corpus("mtg"), first_block * BLOCK, total)
    if kind == "c5":
        mtg = torch.from_numpy(np.frombuffer(fuzzgen.corpus("mtg"), dtype=np.uint8).copy()).to(device)
        span = mtg.numel() - 4096
        k = torch.arange(first_block, first_block + n_blocks, device=device, dtype=torch.int64)
        starts = (k * 4099) % span
        idx = starts[:, None] + torch.arange(4096, device=device, dtype=torch.int64)[None, :]
        pats = mtg[idx]  # [n_blocks, 4096]
        return pats.repeat(1, 16).reshape(-1).contiguous()
    if kind == "mix":  # config C4's recommended mix: block k -> k%3 in {c3, text, c5}
        parts = [make_input(w, first_block, n_blocks, device).reshape(n_blocks, BLOCK) for w in ("c3", "text", "c5")]
        sel = (torch.arange(first_block, first_block + n_blocks, device=device) % 3)
        out = parts[0].clone()
        for j in (1, 2):
            out[sel == j] = parts[j][sel == j]
        return out.reshape(-1).contiguous()
    raise KeyError(kind)


def run_workload(P, kind, first_block, n_blocks, steps, warmup, device, dist_on, verify_blocks=8, decompress=False):
    """Returns dict with timing of `steps` steps (max over ranks), kernel-only event times, sizes."""
    import torch
    import torch.distributed as dist
    import oracle_lib
    from compression_accelerator_amd import sharding as shard

    src = make_input(kind, first_block, n_blocks, device)
    dst = torch.empty(n_blocks * P.SLOT_STRIDE, dtype=torch.uint8, device=device)
    world = dist.get_world_size() if dist_on else 1
    n_total = n_blocks * world
    # Two sets of length buffers: the all_gather + scan of step i run on a side stream under the
    # compression of step i+1 (the collective is latency-bound: 128 KiB per rank).
    out_lens = [torch.zeros(n_blocks, dtype=torch.int32, device=device) for _ in range(2)]
    out_len = out_lens[0]
    if dist_on:
        comm = torch.cuda.Stream(device)
        gathered = [torch.empty(n_total, dtype=torch.int32, device=device) for _ in range(2)]
        offs = [torch.empty(n_total + 1, dtype=torch.int64, device=device) for _ in range(2)]
    consumed = [None, None]

    def step(i, e0=None, e1=None):
        k = i & 1
        main = torch.cuda.current_stream()
        if consumed[k] is not None:
            main.wait_event(consumed[k])  # the gather of step i-2 has read out_lens[k]
        if e0 is not None:
            e0.record()  # same stream as the kernel launch (torch's current stream)
        P.compress_blocks(src, BLOCK, n_blocks, dst=dst, out_len=out_lens[k], varint=True)
        if e1 is not None:
            e1.record()
        if dist_on:
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(comm):
                comm.wait_event(ready)
                shard.gather_lengths(out_lens[k], n_total, out=gathered[k], offsets=offs[k])
                consumed[k] = torch.cuda.Event()
                consumed[k].record(comm)

    for i in range(warmup):
        step(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i, ev[i][0], ev[i][1])
    torch.cuda.synchronize()  # all streams of the device: the last gather too
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the gathered layout is what every rank would pack with: ends at the total, and this rank's
        # slice of it is this rank's own lengths (reported, not asserted: a bench line beats a traceback)
        kk = (warmup + steps - 1) & 1
        r = dist.get_rank()
        layout_ok = (int(offs[kk][-1].item()) == int(gathered[kk].sum().item())
                     and bool(torch.equal(gathered[kk][r * n_blocks:(r + 1) * n_blocks], out_lens[kk])))
    else:
        layout_ok = None
    out_len = out_lens[(warmup + steps - 1) & 1] if steps + warmup > 0 else out_lens[0]
    kernel_ms = [a.elapsed_time(b) for a, b in ev]

    # ---- checks, outside the timed region: sample blocks against the oracle, all lengths sane
    lens = out_len.cpu().numpy().astype(np.int64)
    out_bytes = int(lens.sum())
    pick = np.unique(np.linspace(0, n_blocks - 1, verify_blocks).astype(np.int64))
    host_in = src.reshape(n_blocks, BLOCK)[torch.from_numpy(pick).to(device)].cpu().numpy()
    host_out = dst.reshape(n_blocks, P.SLOT_STRIDE)[torch.from_numpy(pick).to(device)].cpu().numpy()
    exact = True
    for j, b in enumerate(pick):
        want = oracle_lib.raw_compress(host_in[j])
        exact &= host_out[j, :lens[b]].tobytes() == want
    # ---- the next row: GPU decompression of what was just produced, round trip checked
    dec = None
    if decompress:
        back = torch.empty(n_blocks * BLOCK, dtype=torch.uint8, device=device)
        P.uncompress_blocks(dst, P.SLOT_STRIDE, out_len, n_blocks, dst=back)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _, ulen, status = P.uncompress_blocks(dst, P.SLOT_STRIDE, out_len, n_blocks, dst=back)
        e1.record()
        torch.cuda.synchronize()
        dec = {"ms": e0.elapsed_time(e1), "round_trip": bool(torch.equal(back, src)) and int(status.abs().sum().item()) == 0}
        del back
    del src, dst
    return {"elapsed_s": elapsed, "kernel_ms": kernel_ms, "in_bytes": n_blocks * BLOCK, "out_bytes": out_bytes,
            "bit_exact_sample": bool(exact), "n_blocks": n_blocks, "decompress": dec, "gathered_layout_ok": layout_ok}


def cpu_baseline(kind, sample_blocks, budget_s=12.0):
    """The oracle (kind "port": bit-exact CPU restatement of Snappy 1.1.8) on the host cores,
    bounded sample of the same workload."""
    import torch
    import oracle_lib
    src = make_input(kind, 0, sample_blocks, "cpu").numpy()
    cores = min(len(os.sched_getaffinity(0)), 16)  # the box's CPU share for one GPU
    res = {}
    for threads in (1, cores):
        n_pass, t_used = 0, 0.0
        while t_used < budget_s / 2 and n_pass < 50:
            t0 = time.perf_counter()
            oracle_lib.compress_blocks(src, BLOCK, sample_blocks, per_block_varint=True, threads=threads)
            t_used += time.perf_counter() - t0
            n_pass += 1
        res[threads] = sample_blocks * BLOCK * n_pass / t_used / 1e9
    out = {"value": round(res[cores], 3), "unit": "GB/s", "cores": cores, "kind": "port",
           "single_thread_GBps": round(res[1], 3),
           "sample": "%d x 64 KiB blocks of workload %s, oracle/snappy_oracle.c, %d pthreads" % (sample_blocks, kind, cores)}
    lib = _system_libsnappy()
    if lib is not None:  # informational: the real google/snappy of this image, if there is one
        out["system_libsnappy"] = _time_libsnappy(lib, src, sample_blocks, cores, budget_s / 2)
    return out


def _system_libsnappy():
    import ctypes
    for path in ("/opt/conda/lib/libsnappy.so.1", "libsnappy.so.1"):
        try:
            lib = ctypes.CDLL(path)
            lib.snappy_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
            lib.snappy_max_compressed_length.restype = ctypes.c_size_t
            lib.snappy_max_compressed_length.argtypes = [ctypes.c_size_t]
            lib._path = path
            return lib
        except (OSError, AttributeError):
            continue
    return None


def _time_libsnappy(lib, src, sample_blocks, cores, budget_s):
    import ctypes
    from concurrent.futures import ThreadPoolExecutor
    cap = lib.snappy_max_compressed_length(BLOCK)
    base = src.ctypes.data

    def work(rng):
        buf = ctypes.create_string_buffer(cap)
        n = ctypes.c_size_t(0)
        for b in rng:
            n.value = cap
            lib.snappy_compress(base + b * BLOCK, BLOCK, buf, ctypes.byref(n))

    res = {}
    for threads in (1, cores):
        chunks = [range(t, sample_blocks, threads) for t in range(threads)]
        n_pass, t_used = 0, 0.0
        with ThreadPoolExecutor(threads) as ex:
            while t_used < budget_s / 2 and n_pass < 50:
                t0 = time.perf_counter()
                list(ex.map(work, chunks))
                t_used += time.perf_counter() - t0
                n_pass += 1
        res[threads] = sample_blocks * BLOCK * n_pass / t_used / 1e9
    return {"library": lib._path, "single_thread_GBps": round(res[1], 3), "threads": cores,
            "threads_GBps": round(res[cores], 3), "note": "version of the image's libsnappy; bytes equal the oracle's only for 1.1.8"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c3", "text", "c5", "mix"])
    ap.add_argument("--blocks-per-gpu", type=int, default=32768)
    ap.add_argument("--no-extra", action="store_true", help="skip the short text/c5/mix passes")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (nccl = RCCL) even with one rank: rehearsal of the N>1 path")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pkg_loader import load_package
    P = load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.force_dist  # (--force-dist: rehearse the RCCL path with a single rank)
    if dist_on:
        dist.init_process_group("nccl", device_id=device)
    if args.gpus != world and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    B = args.blocks_per_gpu
    first = rank * B
    main_res = run_workload(P, args.workload, first, B, args.steps, args.warmup, device, dist_on)

    extras = {}
    if not args.no_extra:
        for kind, nb in (("text", 16384), ("c5", 16384), ("mix", 16386)):
            if kind == args.workload:
                continue
            r = run_workload(P, kind, rank * nb, nb, 3, 1, device, dist_on, decompress=True)
            k_ms = float(np.mean(r["kernel_ms"]))
            extras[kind] = {"input_GBps": round(r["in_bytes"] * world * 3 / r["elapsed_s"] / 1e9, 2),
                            "kernel_ms": round(k_ms, 3), "blocks_per_gpu": nb,
                            "compression_ratio": round(r["in_bytes"] / r["out_bytes"], 4),
                            "hbm_frac_in_plus_out": round((r["in_bytes"] + r["out_bytes"]) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                            "bit_exact_sample": r["bit_exact_sample"],
                            "gpu_decompress_GBps": round(r["in_bytes"] / (r["decompress"]["ms"] * 1e-3) / 1e9, 2),
                            "gpu_round_trip": r["decompress"]["round_trip"]}

    exact_all = main_res["bit_exact_sample"]
    if dist_on:
        t = torch.tensor([1 if exact_all else 0], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        exact_all = bool(t.item())

    if rank == 0:
        K = args.steps
        total_in = main_res["in_bytes"] * world
        value = total_in * K / main_res["elapsed_s"] / 1e9
        k_ms = float(np.mean(main_res["kernel_ms"]))
        algo_bytes = main_res["in_bytes"] + main_res["out_bytes"] + 4 * B  # per launch (this rank)
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        traffic = None  # HBM bytes per launch from the committed PMC passes (profiles/pmc_summary.py)
        pmc_path = os.path.join(ROOT, "profiles", "pmc_%s_%d.json" % (args.workload, B))
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                traffic = json.load(f).get("hbm_traffic_bytes_per_launch")
        line = {
            "metric": "snappy_compress_input_GBps_64KiB_blocks",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(main_res["elapsed_s"] / K * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": {"c3": "C3 tiled data/randomASCII.txt", "text": "C2-like tiled all-mtg-cards.txt",
                                    "c5": "C5 4 KiB text pattern x16", "mix": "C4 mix c3/text/c5"}[args.workload]
                       + ", %d x 64 KiB blocks per GPU (%.2f GiB/GPU)" % (B, B * BLOCK / 2**30),
                       "blocks_per_gpu": B, "block_bytes": BLOCK,
                       "kernel": os.environ.get("SZ_KERNEL", "auto (sampled: single for incompressible, dual otherwise)"),
                       "sharding": "contiguous block ranges, no data-path collective; all_gather of uint32 lengths over RCCL"},
            "compressed_size_ratio_vs_ref": 1.0 if exact_all else None,
            "bit_exact_vs_oracle_sample": exact_all,
            "compression_ratio": round(main_res["in_bytes"] / main_res["out_bytes"], 5),
            "input_frac_of_hbm_peak": round(main_res["in_bytes"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": int(traffic) if traffic else None,
                         "kernel": "szk::compress_blocks_kernel", "kernel_ms_avg": round(k_ms, 4),
                         "algorithmic_bytes_per_launch": algo_bytes},
            "workloads": extras,
        }
        if main_res.get("gathered_layout_ok") is not None:
            line["gathered_layout_ok"] = main_res["gathered_layout_ok"]  # rank 0's view of the all_gather result
        if not args.no_cpu and world == 1:  # host baseline: rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(args.workload, 1024)
            if "text" in extras:  # the same oracle on the latency-bound workload, for scale (bounded: ~4 s)
                t = cpu_baseline("text", 256, budget_s=4.0)
                extras["text"]["cpu_port_GBps"] = {"1_thread": t["single_thread_GBps"], "%d_threads" % t["cores"]: t["value"]}
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
