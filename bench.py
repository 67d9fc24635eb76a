#!/usr/bin/env python3
"""bench.py -- Snappy block compression throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|text|c5|mix] [--blocks-per-gpu B]

A "step" is one pass of the hot path (sz_compress_blocks_hip through the C ABI) over this rank's
blocks, input already resident in HBM, plus -- for N>1 -- the RCCL all_gather of the per-block
lengths.  Prints ONE JSON line on rank 0.  Blocks shard by contiguous ranges, weak scaling (fixed
blocks per GPU), one process per GPU.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts the N ranks itself
(`python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>`) as a CHILD
process before anything here has touched the GPU, relays rank 0's line and exits with the child's
code.  Under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.

Headline workload (config.workload): C3 = consecutive 64 KiB windows of the infinite tiling of
data/randomASCII.txt (BASELINE.json configs[2], the one the north-star target is quoted on);
--blocks-per-gpu 32768 (2 GiB per GPU) so that N=8 is config C4's 16 GiB.  Short passes over the
text (C2-like), low-entropy (C5) and mixed (C4 mix) workloads are reported in "workloads"; "one_stream" is the
next rows' number (one 1 GiB raw stream compressed + packed, then decoded from its bytes alone), "hw_compat" the
functional model of the reference accelerator's datapath on the text workload.
Every block of every pass is compared with the oracle after the timed region.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

BLOCK = 65536
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
EVENT_EVERY = 4         # timed steps between two that carry timing events (they cost ~1.4 % of the step they are on)


class HipEvents:
    """Raw hipEvent_t pairs for sz_profile_next_launch (include/snappy_hip.h): the library records them on the
    launch's stream right around the kernel that can be a launch's dominant one -- `single` = one wavefront per block,
    `multi` = the persistent kernel for compressible launches; both are enqueued when a sample of the data chooses,
    the one not chosen exits at once.  The runtime is the one the process already has (torch's libamdhip64)."""

    def __init__(self, n):
        import ctypes
        self.ct = ctypes
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.ev = []
        for _ in range(n):
            four = []
            for _ in range(4):
                e = ctypes.c_void_p()
                assert self.hip.hipEventCreate(ctypes.byref(e)) == 0
                four.append(e.value)
            self.ev.append(four)

    def ms(self, i, multi):
        """ms between the pair of step i that was handed to the library (single: 0, 1; multi: 2, 3); None if the kernel
        was not launched (the events were never recorded: the error that leaves behind is taken off the runtime's
        last-error slot, where torch would find it at its next call)."""
        a, b = (2, 3) if multi else (0, 1)
        t = self.ct.c_float()
        rc = self.hip.hipEventElapsedTime(self.ct.byref(t), self.ev[i][a], self.ev[i][b])
        if rc != 0:
            self.hip.hipGetLastError()
            return None
        return float(t.value)

    def close(self):
        for four in self.ev:
            for e in four:
                self.hip.hipEventDestroy(e)
        self.ev = []


def parse_args(argv=None):
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
    return ap.parse_args(argv)


# ------------------------------------------------------------------------- launcher (no GPU call)

def launch_ranks(n, script, script_args, env=None, timeout=None):
    """Starts n ranks of `script` under torch.distributed.run on this node as a child process (never
    an exec: a process that has initialised the GPU must not be replaced, and this one stays clean
    anyway), relays the child's stdout/stderr and returns (exit code, the last JSON line printed).
    The rendezvous is torchrun's own (`--standalone`: a port it binds itself, nothing probed and
    released here); the child is the leader of a new session, so that on a timeout or an error the
    whole group -- the launcher AND its ranks -- is killed and no rank keeps a GPU."""
    import signal
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1",
           "--nnodes=1", "--nproc-per-node", str(n), script] + list(script_args)
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this pool)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)  # (a launcher started from inside a rank must not hand its own identity down)
    p = subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except BaseException:
        try:
            os.killpg(p.pid, signal.SIGKILL)  # the session's process group: launcher and ranks
        except ProcessLookupError:
            pass
        p.wait()
        raise
    if p.returncode != 0:  # a failed rank: torchrun has stopped the others; make sure of it
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
    line = None
    for ln in out.splitlines():
        if ln.startswith("{") and ln.rstrip().endswith("}"):
            line = ln
    return p.returncode, line, out, err


def verify_gather(gathered, offsets, local_lens, rank, world, n_total, group_is_up):
    """What is reported about the one collective of the path -- checked, not inferred from a tensor's size.  Outside the
    timed region every rank publishes a tag of what it HAD (its rank, its number of blocks, the sum and a
    position-weighted checksum of its own lengths; one more tiny all_gather); every rank then compares each tag with the
    slice of that rank in the lengths it RECEIVED in the timed step.  `rccl_ranks_seen` = ranks whose slice arrived as
    sent (must be `world`), `gathered_layout_ok` = all of them did, the layout ends at the total, and this rank's
    slice is its own lengths.  Tensors on any device; with one rank (no process group) the tag is compared locally."""
    import torch
    from compression_accelerator_amd import sharding as shard

    def tag_of(lens, r):
        v = lens.to(torch.int64)
        w = torch.arange(1, v.numel() + 1, dtype=torch.int64, device=v.device)
        return [r, int(v.numel()), int(v.sum().item()), int(((v * w).sum() % 2305843009213693951).item())]

    mine = torch.tensor(tag_of(local_lens, rank), dtype=torch.int64, device=gathered.device)
    if group_is_up and world > 1:
        import torch.distributed as dist
        tags = torch.empty(4 * world, dtype=torch.int64, device=gathered.device)
        dist.all_gather_into_tensor(tags, mine)
        tags = tags.reshape(world, 4).cpu().tolist()
    else:
        tags = [mine.cpu().tolist()]
    seen = 0
    for r in range(world):
        b, e = shard.shard_range(n_total, r, world)
        seen += 1 if tags[r] == tag_of(gathered[b:e], r) else 0
    b, e = shard.shard_range(n_total, rank, world)
    ok = (seen == world and int(gathered.numel()) == n_total
          and int(offsets[-1].item()) == int(gathered.to(torch.int64).sum().item())
          and bool(torch.equal(gathered[b:e], local_lens)))
    return {"rccl_ranks_seen": seen, "gathered_layout_ok": ok}


# ------------------------------------------------------------------------------- inputs (on device)

def make_input(kind, first_block, n_blocks, device):
    """Builds this rank's blocks [first_block, first_block+n_blocks) directly in HBM."""
    import numpy as np
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


# ------------------------------------------------------------------------------------ one workload

def run_workload(P, kind, first_block, n_blocks, steps, warmup, device, dist_on, decompress=False):
    """Returns dict with timing of `steps` steps (max over ranks), kernel-only event times, sizes."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import parity_util
    from compression_accelerator_amd import sharding as shard

    src = make_input(kind, first_block, n_blocks, device)
    dst = torch.empty(n_blocks * P.SLOT_STRIDE, dtype=torch.uint8, device=device)
    world = dist.get_world_size() if dist_on else 1
    n_total = n_blocks * world
    # Two sets of length buffers: the all_gather + scan of step i run on a side stream under the
    # compression of step i+1 (the collective is latency-bound: 128 KiB per rank).
    out_lens = [torch.zeros(n_blocks, dtype=torch.int32, device=device) for _ in range(2)]
    if dist_on:
        comm = torch.cuda.Stream(device)
        gathered = [torch.empty(n_total, dtype=torch.int32, device=device) for _ in range(2)]
        offs = [torch.empty(n_total + 1, dtype=torch.int64, device=device) for _ in range(2)]
    consumed = [None, None]
    hev = None
    # (which kernel does a workload's work is the library's choice -- per launch from a sample, per block in a mixed
    # launch --: all four events go to every sampled step and the longer pair is the dominant kernel)

    def step(i, e0=None, e1=None):
        k = i & 1
        main = torch.cuda.current_stream()
        if consumed[k] is not None:
            main.wait_event(consumed[k])  # the gather of step i-2 has read out_lens[k]
        # Events only on every 4th timed step: a timed pair around the call costs 0.7 % of an incompressible 2 GiB
        # step and a pair around the kernel another 0.7 % (tools/evcost.py), and `value` is the rate without them.
        sampled_step = e0 is not None and (i - warmup) % EVENT_EVERY == 0
        if sampled_step:
            e0.record()  # same stream as the kernel launch (torch's current stream)
            four = hev.ev[i - warmup]
            P.profile_next_launch(four[0], four[1], four[2], four[3])  # events around both compress kernels, recorded by the library
        P.compress_blocks(src, BLOCK, n_blocks, dst=dst, out_len=out_lens[k], varint=True)
        if sampled_step:
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
    hev = HipEvents(steps)
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
    last = (warmup + steps - 1) & 1 if steps + warmup > 0 else 0
    layout = None
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # (reported, not asserted: a bench line beats a traceback)
        layout = verify_gather(gathered[last], offs[last], out_lens[last], dist.get_rank(), world, n_total, True)
    out_len = out_lens[last]
    sampled = [i for i in range(steps) if i % EVENT_EVERY == 0]
    call_ms = [ev[i][0].elapsed_time(ev[i][1]) for i in sampled]  # the whole call: sample + both compress kernels (one of them exits at once unless the launch is mixed)
    # the dominant kernel alone (HIP events recorded by the library around its launch, on the launch's stream)
    single_ms = [hev.ms(i, False) for i in sampled]
    multi_ms = [hev.ms(i, True) for i in sampled]
    hev.close()
    mean_of = lambda v: (sum(x for x in v if x is not None) / max(1, sum(1 for x in v if x is not None))) if any(x is not None for x in v) else 0.0
    single_is_dominant = mean_of(single_ms) >= mean_of(multi_ms)
    kernel_name = "szk::compress_blocks_kernel" if single_is_dominant else (
        "szk::compress_blocks_team_kernel" if os.environ.get("SZ_KERNEL") != "dual" else "szk::compress_blocks_dual_kernel")
    kernel_ms = []
    for j in range(len(sampled)):
        k_ms = single_ms[j] if single_is_dominant else multi_ms[j]
        kernel_ms.append(k_ms if k_ms is not None else call_ms[j])  # (no kernel recorded its events: the call)
    other_ms = mean_of(multi_ms if single_is_dominant else single_ms)  # the other compress kernel of the call (it exits at once, or takes its blocks of a mixed launch)

    # ---- checks, outside the timed region: EVERY block against the oracle (bytes and lengths)
    lens = out_len.cpu().numpy().astype(np.int64)
    out_bytes = int(lens.sum())
    bad, first_bad = parity_util.count_mismatching_blocks(src, dst, out_len, n_blocks, P.SLOT_STRIDE)
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
    return {"elapsed_s": elapsed, "kernel_ms": kernel_ms, "call_ms": call_ms, "kernel_name": kernel_name, "other_kernel_ms": other_ms,
            "in_bytes": n_blocks * BLOCK, "out_bytes": out_bytes,
            "blocks_differing_from_oracle": bad, "first_bad_block": first_bad, "n_blocks": n_blocks,
            "decompress": dec, "layout": layout}


def one_stream(P, kind, n_blocks, device, reps=3):
    """The rows next to the hot path on ONE raw stream resident in HBM: fragments compressed and
    packed into `varint32(n) + bodies` on the device, then decoded from nothing but those bytes (fragment offsets
    found by the stream index).  Kernel-side times (events on the launch stream), best of `reps`; round trip checked."""
    import torch
    src = make_input(kind, 0, n_blocks, device)
    n = n_blocks * BLOCK
    hdr, v = [], n
    while v >= 0x80:
        hdr.append(0x80 | (v & 0x7f))
        v >>= 7
    hdr.append(v)
    slots = torch.empty(n_blocks * P.SLOT_STRIDE, dtype=torch.uint8, device=device)
    lens = torch.empty(n_blocks, dtype=torch.int32, device=device)
    packed = torch.empty(len(hdr) + n_blocks * P.SLOT_STRIDE + 64, dtype=torch.uint8, device=device)
    offs = torch.empty(n_blocks + 1, dtype=torch.int64, device=device)
    back = torch.empty(n, dtype=torch.uint8, device=device)
    packed[:len(hdr)] = torch.tensor(hdr, dtype=torch.uint8, device=device)
    t_c, t_d, st = [], [], None
    for i in range(reps + 1):
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record()
        P.compress_stream(src, n, slots=slots, lens=lens)
        P.pack_stream(slots, P.SLOT_STRIDE, lens, n_blocks, base=len(hdr), packed=packed, offsets=offs)
        b.record()
        total = int(offs[n_blocks].item())
        b2 = torch.cuda.Event(enable_timing=True)
        b2.record()
        _, st = P.uncompress_stream(packed, total, len(hdr), n, dst=back)
        c.record()
        torch.cuda.synchronize()
        if i:  # (the first pass warms up: workspaces, code objects)
            t_c.append(a.elapsed_time(b))
            t_d.append(b2.elapsed_time(c))
    ok = st.cpu().numpy().tolist() == [0, 0, 0] and bool(torch.equal(back, src))
    return {"workload": kind, "uncompressed_bytes": n, "stream_bytes": total,
            "compress_and_pack_GBps": round(n / (min(t_c) * 1e-3) / 1e9, 2),
            "indexed_decode_GBps": round(n / (min(t_d) * 1e-3) / 1e9, 2), "round_trip": ok}


def hw_compat_row(P, kind, n_blocks, device, reps=3):
    """the same blocks through SZ_FLAG_HW_COMPAT -- the element stream of a functional model of the
    reference accelerator's datapath (size-exact on 31 of the 36 rows of benchmark/hw_results.csv, tests/test_hw_model.py;
    a compatibility mode, one wave per block) -- rate, ratio, and that the software decoder reads the slots back."""
    import torch
    src = make_input(kind, 0, n_blocks, device)
    dst, ol = P.compress_blocks(src, BLOCK, n_blocks, hw_compat=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        P.compress_blocks(src, BLOCK, n_blocks, dst=dst, out_len=ol, hw_compat=True)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    back, _, st = P.uncompress_blocks(dst, P.SLOT_STRIDE, ol, n_blocks)
    ok = bool(torch.equal(back, src)) and int(st.abs().sum().item()) == 0
    return {"workload": kind, "blocks": n_blocks, "input_GBps": round(n_blocks * BLOCK / (min(ts) * 1e-3) / 1e9, 2),
            "compression_ratio": round(n_blocks * BLOCK / float(ol.sum().item()), 4), "decodes_back": ok,
            "what": "functional model of the reference accelerator's datapath (size-exact on 31 of 36 hw_results.csv rows)"}


def host_api_rates(P, n_bytes, reps=3):
    """The reference's own call surface (snappy::Compress / RawCompress / Uncompress on host memory: tests/swcompression.cc:45,
    workloads/snappy/overlay/root/compress.cc:17 -> sz_compress / sz_uncompress), PCIe included, on pageable host buffers
    of n_bytes: median wall time of `reps` calls after one warm-up call (which allocates the thread's workspace).  Never
    `value`: the bench's metric is the kernel path with its input resident in HBM.  Bytes checked against the oracle."""
    import ctypes
    import numpy as np
    import oracle_lib
    L = P.lib()
    out = {"bytes": n_bytes, "calls": reps,
           "what": "sz_compress / sz_uncompress on pageable host memory, wall clock, PCIe both ways included: a pipeline of 64..256 MiB "
                   "stages (copy in | compress + pack | copy out) with the caller's buffers registered for DMA for the length of the call"}
    cap = P.max_compressed_length(n_bytes)
    # (written once before the calls, as a caller's buffers are -- snappy::Compress resizes its std::string, which zero-fills
    # it; pages that no one has touched yet are faulted in one by one under the first copy that lands on them)
    dst = np.zeros(cap, dtype=np.uint8)
    back = np.zeros(n_bytes, dtype=np.uint8)
    dst[::4096] = 1
    back[::4096] = 1
    got = ctypes.c_size_t(0)
    for kind, name in (("c3", "incompressible"), ("text", "text")):
        src = make_input(kind, 0, n_bytes // BLOCK, "cpu").numpy()
        ts = []
        for i in range(reps + 1):
            t0 = time.perf_counter()
            rc = L.sz_compress(src.ctypes.data, n_bytes, dst.ctypes.data, cap, ctypes.byref(got))
            ts.append(time.perf_counter() - t0)
            assert rc == 0, rc
        clen = got.value
        tu = []
        for i in range(reps + 1):
            t0 = time.perf_counter()
            rc = L.sz_uncompress(dst.ctypes.data, clen, back.ctypes.data, n_bytes, ctypes.byref(got))
            tu.append(time.perf_counter() - t0)
            assert rc == 0, rc
        want = oracle_lib.raw_compress(src)
        out[name] = {"compress_ms": [round(t * 1e3, 1) for t in ts], "uncompress_ms": [round(t * 1e3, 1) for t in tu],
                     "compress_GBps": round(n_bytes / sorted(ts[1:])[len(ts[1:]) // 2] / 1e9, 2),
                     "uncompress_GBps": round(n_bytes / sorted(tu[1:])[len(tu[1:]) // 2] / 1e9, 2),
                     "first_call_ms": round(ts[0] * 1e3, 1), "stream_bytes": clen,
                     "bytes_equal_oracle": bool(clen == len(want) and dst[:clen].tobytes() == want),
                     "round_trip": bool(np.array_equal(back, src))}
        del src, want
    L.sz_release_thread_resources()
    return out


def copy_roof(n_bytes, device, reps=10):
    """A plain device-to-device copy of as many bytes as the headline workload reads (torch's copy kernel, the
    same process, the same box, HIP events on the stream it runs on): what this part moves when a kernel
    does nothing but copy.  Returns in+out GB/s (best and mean of `reps` after two warm-ups)."""
    import torch
    a = torch.empty(n_bytes // 8, dtype=torch.int64, device=device)
    a.random_(0, 1 << 40)
    b = torch.empty_like(a)
    res = {}
    # two forms of "copy": the runtime's buffer copy (Tensor.copy_) and a generic elementwise kernel that reads and
    # writes every element once (x + 0 on 8-byte elements); the faster one is the roof
    for name, fn in (("copy_", lambda: b.copy_(a)), ("add_out", lambda: torch.add(a, 0, out=b))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ms = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        res[name] = (2.0 * n_bytes / (min(ms) * 1e-3) / 1e9, 2.0 * n_bytes / (sum(ms) / len(ms) * 1e-3) / 1e9)
    del a, b
    best = max(res, key=lambda k: res[k][0])
    return {"best": res[best][0], "mean": res[best][1], "form": best,
            "all": {k: round(v[0], 1) for k, v in res.items()}}


# -------------------------------------------------------------------------------------- CPU baseline

def effective_cores():
    """CPUs this process may really use: the affinity set, capped by the cgroup CPU quota (a GPU box
    shows all of the host's logical CPUs in the affinity mask while the container owns a share)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def host_cpu():
    model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    cores, quota = effective_cores()
    return {"model": model, "os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)),
            "cgroup_cpu_quota": quota, "threads_used": cores}


CPU_BATCHES = 5  # timed batches per thread count: the reported rate is their median, the spread is reported beside it


def _timed_batches(time_fn, sample_blocks, threads, budget_s):
    """CPU_BATCHES timed batches of `reps` passes each of time_fn(threads, reps) -> seconds (one untimed probe pass sizes
    `reps` so that all batches together take about budget_s).  Returns the per-batch rates in GB/s and reps.  A shared
    box moves a single batch by several x (round 5's driver line had the image's libsnappy at 1.83 GB/s on one thread,
    13.2 elsewhere): the median of five batches with min / max beside it says whether a figure can be trusted."""
    sec1 = max(time_fn(threads, 1), 1e-6)
    reps = max(1, min(400, int(budget_s / CPU_BATCHES / sec1)))
    rates = []
    for _ in range(CPU_BATCHES):
        sec = time_fn(threads, reps)
        rates.append(sample_blocks * BLOCK * reps / sec / 1e9)
    return rates, reps


def _spread(rates):
    r = sorted(rates)
    return {"median": round(r[len(r) // 2], 3), "min": round(r[0], 3), "max": round(r[-1], 3), "batches": len(r),
            "noisy": bool(r[-1] > 1.5 * r[0])}


def cpu_baseline(kind, sample_blocks, budget_s=10.0):
    """The oracle (kind "port": bit-exact CPU restatement of Snappy 1.1.8) on the host cores, on a
    bounded sample of the same workload: pinned pthreads created once, buffers allocated before the
    clock starts, one warm-up pass, then CPU_BATCHES timed batches per thread count (oracle_time_compress_blocks; the
    reference's own loop is workloads/snappy/overlay/root/compress.cc:26-34).  `value` = the MEDIAN batch on all
    granted cores; `spread` carries min / max and `noisy` (max > 1.5 x min: another tenant had the cores)."""
    import oracle_lib
    src = make_input(kind, 0, sample_blocks, "cpu").numpy()
    cores, _ = effective_cores()  # the box's real CPU share
    cores = max(1, min(cores, sample_blocks // 8))  # (every thread gets a few blocks)
    res = {}
    for threads in (1, cores):
        rates, reps = _timed_batches(lambda t, r: oracle_lib.time_compress_blocks(src, BLOCK, sample_blocks, t, r)[0],
                                     sample_blocks, threads, budget_s / 2)
        res[threads] = (_spread(rates), reps)
    out = {"value": res[cores][0]["median"], "median": res[cores][0]["median"], "unit": "GB/s", "cores": cores, "kind": "port",
           "spread": res[cores][0], "noisy": bool(res[cores][0]["noisy"] or res[1][0]["noisy"]),
           "single_thread_GBps": res[1][0]["median"], "single_thread_spread": res[1][0],
           "passes_per_batch": {"1_thread": res[1][1], "%d_threads" % cores: res[cores][1]},
           "host": host_cpu(),
           "sample": "%d x 64 KiB blocks of workload %s, oracle/snappy_oracle.c, %d pinned pthreads, preallocated buffers, "
                     "1 warm-up pass + %d timed batches per thread count (median reported)" % (sample_blocks, kind, cores, CPU_BATCHES)}
    lib = _system_libsnappy()
    if lib is not None:  # informational: the real google/snappy of this image, if there is one
        out["system_libsnappy"] = _time_libsnappy(lib, src, sample_blocks, cores, budget_s / 2)
        out["noisy"] = bool(out["noisy"] or out["system_libsnappy"]["noisy"])
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
    """The image's google/snappy (C API) in the SAME harness as the port: pinned pthreads created once,
    preallocated buffers, one warm-up pass, CPU_BATCHES timed batches per thread count, median reported
    (oracle_time_external_blocks calls the library's snappy_compress per 64 KiB block)."""
    import ctypes
    import oracle_lib
    fn = ctypes.cast(lib.snappy_compress, ctypes.c_void_p).value
    res = {}
    for threads in (1, cores):
        rates, _ = _timed_batches(lambda t, r: oracle_lib.time_external_blocks(fn, src, BLOCK, sample_blocks, t, r)[0],
                                  sample_blocks, threads, budget_s / 2)
        res[threads] = _spread(rates)
    return {"library": lib._path, "single_thread_GBps": res[1]["median"], "single_thread_spread": res[1], "threads": cores,
            "threads_GBps": res[cores]["median"], "threads_spread": res[cores],
            "noisy": bool(res[1]["noisy"] or res[cores]["noisy"]),
            "note": "the image's libsnappy, snappy_compress per block, same pinned-thread harness as the port; "
                    "its bytes equal the oracle's only if it is 1.1.8"}


def load_pmc(kind, n_blocks, library_version=""):
    """The committed summary of the rocprofv3 --pmc passes of this workload at this size (profiles/pmc_<kind>_<blocks>.json,
    written by tools/pmc_r6_sum.py -- round 5's files by tools/pmc_r5_sum.py -- from the passes tools/profile_r06.sh
    runs), or {}.  PMC counters need the profiler, so they are not re-measured in a bench run; the file says which
    sources the measured library was built from (`library_source_id`, the `src` field of sz_version()) and at which
    commit it was collected, and `_stale` is true when that is not the library this run loaded."""
    path = os.path.join(ROOT, "profiles", "pmc_%s_%d.json" % (kind, n_blocks))
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        d = json.load(f)
    sid = d.get("library_source_id")
    d["_stale"] = (sid is None) or (("src " + sid) not in library_version)
    d["_source"] = ("profiles/%s (rocprofv3 --pmc passes, committed; not re-measured in this run; collected at commit %s with the "
                    "library built from sources %s: %s)" % (os.path.basename(path), d.get("collected_at_commit", "?"), sid or "?",
                                                            "an OLDER build than the one this run loaded" if d["_stale"] else "the build this run loaded"))
    return d


# ----------------------------------------------------------------------------------------- one rank

def rank_main(args):
    # ONE line on stdout: the JSON.  Libraries that write to file descriptor 1 on their own (RCCL prints a five-line
    # version banner at its first collective) are sent to stderr; the line itself goes to the descriptor kept here.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import numpy as np
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
    props = torch.cuda.get_device_properties(device)
    if "gfx950" not in getattr(props, "gcnArchName", "gfx950") and rank == 0:
        print("note: device %s is not gfx950; the roofline peak below is MI355X's" % props.name, file=sys.stderr)

    lib_version = P.lib().sz_version().decode()  # names the toolchain ("untuned toolchain" if not the tuned one) and the sources (`src <id>`)
    B = args.blocks_per_gpu
    main_res = run_workload(P, args.workload, rank * B, B, args.steps, args.warmup, device, dist_on)

    extras = {}
    if not args.no_extra:
        for kind, nb in (("text", 16384), ("c5", 16384), ("mix", 16386)):
            if kind == args.workload:
                continue
            xsteps = 10
            r = run_workload(P, kind, rank * nb, nb, xsteps, 2, device, dist_on, decompress=True)
            k_ms = float(np.mean(r["kernel_ms"]))
            algo = r["in_bytes"] + r["out_bytes"] + 4 * nb  # algorithmic bytes per launch: input read once + output written once + lengths
            pmc_w = load_pmc(kind, nb, lib_version)
            extras[kind] = {"input_GBps": round(r["in_bytes"] * world * xsteps / r["elapsed_s"] / 1e9, 2), "steps": xsteps,
                            "kernel_ms": round(k_ms, 3), "kernel": r["kernel_name"], "blocks_per_gpu": nb,
                            "compression_ratio": round(r["in_bytes"] / r["out_bytes"], 4),
                            "hbm_frac_in_plus_out": round((r["in_bytes"] + r["out_bytes"]) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                            "roofline": {"bound": "hbm", "kernel": r["kernel_name"], "kernel_ms_avg": round(k_ms, 4),
                                         "other_compress_kernel_ms_avg": round(r["other_kernel_ms"], 4),
                                         "call_ms_avg": round(float(np.mean(r["call_ms"])), 4),
                                         "algorithmic_bytes_per_launch": algo,
                                         "achieved": round(algo / (k_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                         "frac": round(algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                         "traffic": int(pmc_w["hbm_traffic_bytes_per_launch"]) if pmc_w.get("hbm_traffic_bytes_per_launch") else None,
                                         "traffic_source": pmc_w.get("_source"), "traffic_stale": pmc_w.get("_stale"),
                                         "lds_bank_conflict_frac": pmc_w.get("lds_bank_conflict_frac"),
                                         "resident_waves_per_simd": pmc_w.get("resident_waves_per_simd")},
                            "blocks_differing_from_oracle": r["blocks_differing_from_oracle"],
                            "gpu_decompress_GBps": round(r["in_bytes"] / (r["decompress"]["ms"] * 1e-3) / 1e9, 2),
                            "gpu_round_trip": r["decompress"]["round_trip"]}
            if r["layout"] is not None:
                extras[kind].update(r["layout"])

    # The same workloads at four times the size (N = 1 only, 5 steps; `workloads` above stays the row of record): a
    # 1 GiB launch of text ends ~1.4 ms after its queue is empty -- whole blocks of 2-4 ms on the last workers --, which is
    # 9 % of it and 2 % of a 4 GiB one (EXPERIMENTS.md R5.4, last item); every block compared with the oracle as everywhere.
    large = {}
    if not args.no_extra and world == 1:
        for kind, nb in (("text", 65536), ("mix", 65538)):
            r = run_workload(P, kind, 0, nb, 5, 1, device, False)
            large[kind] = {"input_GBps": round(r["in_bytes"] * 5 / r["elapsed_s"] / 1e9, 2), "blocks_per_gpu": nb, "steps": 5,
                           "kernel_ms": round(float(np.mean(r["kernel_ms"])), 3), "kernel": r["kernel_name"],
                           "blocks_differing_from_oracle": r["blocks_differing_from_oracle"]}

    stream_row = hw_row = host_row = None
    if not args.no_extra and rank == 0 and world == 1:
        stream_row = one_stream(P, "text", 16384, device)
        hw_row = hw_compat_row(P, "text", 16384, device)  # (two rounds of its 8 192 wave slots)
        torch.cuda.empty_cache()
        host_row = host_api_rates(P, 2 << 30)  # (tools/hostrate.py: the same size through the snappy:: facade, both call shapes)

    bad_all = (main_res["blocks_differing_from_oracle"] + sum(v["blocks_differing_from_oracle"] for v in extras.values())
               + sum(v["blocks_differing_from_oracle"] for v in large.values()))
    if dist_on:
        t = torch.tensor([bad_all], device=device, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        bad_all = int(t.item())

    if rank == 0:
        K = args.steps
        total_in = main_res["in_bytes"] * world
        value = total_in * K / main_res["elapsed_s"] / 1e9
        k_ms = float(np.mean(main_res["kernel_ms"]))
        algo_bytes = main_res["in_bytes"] + main_res["out_bytes"] + 4 * B  # per launch (this rank)
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        pmc = load_pmc(args.workload, B, lib_version)  # HBM bytes per launch etc. from the committed PMC passes
        traffic = pmc.get("hbm_traffic_bytes_per_launch")
        roof = copy_roof(main_res["in_bytes"], device) if world == 1 else None
        line = {
            "metric": "snappy_compress_input_GBps_64KiB_blocks",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(main_res["elapsed_s"] / K * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "library": lib_version,
            "config": {"workload": {"c3": "C3 tiled data/randomASCII.txt", "text": "C2-like tiled all-mtg-cards.txt",
                                    "c5": "C5 4 KiB text pattern x16", "mix": "C4 mix c3/text/c5"}[args.workload]
                       + ", %d x 64 KiB blocks per GPU (%.2f GiB/GPU)" % (B, B * BLOCK / 2**30),
                       "blocks_per_gpu": B, "block_bytes": BLOCK,
                       "kernel": os.environ.get("SZ_KERNEL", "auto (sampled: single kernel for incompressible launches, team kernel otherwise)"),
                       "sharding": "contiguous block ranges, no data-path collective; all_gather of uint32 lengths over RCCL",
                       "device": props.name},
            "compressed_size_ratio_vs_ref": 1.0 if bad_all == 0 else None,
            "bit_exact_vs_oracle_all_blocks": bad_all == 0,
            "blocks_differing_from_oracle": bad_all,
            "compression_ratio": round(main_res["in_bytes"] / main_res["out_bytes"], 5),
            "input_frac_of_hbm_peak": round(main_res["in_bytes"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": int(traffic) if traffic else None,
                         "traffic_source": pmc.get("_source") if traffic else None, "traffic_stale": pmc.get("_stale") if traffic else None,
                         "lds_bank_conflict_frac": pmc.get("lds_bank_conflict_frac"),
                         "resident_waves_per_simd": pmc.get("resident_waves_per_simd"),
                         "copy_roof_GBps": round(roof["best"], 1) if roof else None,
                         "copy_roof_mean_GBps": round(roof["mean"], 1) if roof else None,
                         "frac_of_copy_roof": round(achieved / roof["best"], 4) if roof else None,
                         "copy_roof_what": ("plain copy of the same %d input bytes by torch (%s; in+out bytes / best HIP-event time of 10, same process, "
                                            "same box); forms tried: %s" % (main_res["in_bytes"], roof["form"], roof["all"])) if roof else None,
                         "kernel": main_res["kernel_name"] or "szk::compress_blocks_kernel", "kernel_ms_avg": round(k_ms, 4),
                         "kernel_ms_how": "HIP events recorded by the library right around this kernel's launch on the launch's stream (sz_profile_next_launch), mean over every %dth timed step" % EVENT_EVERY,
                         "call_ms_avg": round(float(np.mean(main_res["call_ms"])), 4),
                         "call_ms_what": "the whole call on the steps that carry events (three timed pairs: ~6 us each); what the call costs beside its dominant kernel without them: side_kernels_us",
                         "side_kernels_us": round((main_res["elapsed_s"] / K * 1e3 - k_ms) * 1e3, 1),
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "peak_source": "MI355X HBM3E spec (MI355X_MICROARCH.md); device reports %s, %d CUs"
                                        % (props.name, props.multi_processor_count)},
            "workloads": extras,
            "workloads_at_4x_size": large,
            "one_stream": stream_row,
            "hw_compat": hw_row,
            "host_api": host_row,
        }
        if main_res["layout"] is not None:
            line.update(main_res["layout"])  # rank 0's view of the all_gather result
        if not args.no_cpu and world == 1:  # host baseline: rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(args.workload, 1024)
            for kind, nb_s, budget in (("text", 256, 6.0), ("c5", 512, 3.0), ("mix", 384, 4.0)):
                if kind in extras:  # the same oracle on the other reported workloads (bounded)
                    t = cpu_baseline(kind, nb_s, budget_s=budget)
                    extras[kind]["cpu_baseline"] = {k: t[k] for k in ("value", "unit", "cores", "kind", "spread", "noisy", "single_thread_GBps", "sample")}
                    if "system_libsnappy" in t:
                        extras[kind]["cpu_baseline"]["system_libsnappy"] = t["system_libsnappy"]
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    # The line is out; a run whose blocks differ from the oracle, or whose length gather did not bring every rank's
    # lengths as sent, FAILS (exit status, after the line: a bench line beats a traceback, but not a green light)
    verdict = 0
    if bad_all != 0:
        verdict = 3
    if dist_on and rank == 0:
        views = [main_res["layout"]] + [{k: v[k] for k in ("rccl_ranks_seen", "gathered_layout_ok")} for v in extras.values() if "rccl_ranks_seen" in v]
        if any(v is None or v["rccl_ranks_seen"] != world or not v["gathered_layout_ok"] for v in views):
            verdict = 4
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if verdict:
        print("bench.py: FAILED (%s)" % ("blocks differ from the oracle" if verdict == 3 else "the length gather did not deliver every rank's lengths"), file=sys.stderr)
        raise SystemExit(verdict)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # start the ranks as a child process; nothing in THIS process has touched the GPU
        rc, line, out, err = launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
        sys.stderr.write(err)
        if line is not None:
            print(line, flush=True)
        else:
            sys.stdout.write(out)
        raise SystemExit(rc)
    rank_main(args)


if __name__ == "__main__":
    main()
