"""world_size-2 gloo test (CPU) of the multi-GPU path: block-range sharding + the one collective
(all_gather of per-block lengths).  Per-rank lengths come from the oracle here; on GPUs they come
from the kernel, the exchange code is the same (compression-accelerator_amd/sharding.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fuzzgen
import oracle_lib
from pkg_loader import load_package


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _blocks(n_total):
    data = fuzzgen.corpus("mtg")
    return [data[k * 65536:(k + 1) * 65536] for k in range(n_total)]


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    load_package()
    from compression_accelerator_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = sharding.shard_range(n_total, rank, world)
    lens = torch.tensor([len(oracle_lib.raw_compress(x)) for x in _blocks(n_total)[b:e]], dtype=torch.int32)
    all_lens, offs = sharding.gather_lengths(lens, n_total)
    # steady-state form (bench.py): preallocated results, reused every step
    out = torch.full((n_total,), -1, dtype=torch.int32)
    off2 = torch.full((n_total + 1,), -1, dtype=torch.int64)
    for _ in range(2):
        a2, o2 = sharding.gather_lengths(lens, n_total, out=out, offsets=off2)
        assert a2.data_ptr() == out.data_ptr() and o2.data_ptr() == off2.data_ptr()
        assert torch.equal(a2, all_lens) and torch.equal(o2, offs)
    q.put((rank, all_lens.numpy().tolist(), offs.numpy().tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7])
def test_two_rank_length_gather_matches_single_process(n_total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = [len(oracle_lib.raw_compress(x)) for x in _blocks(n_total)]
    want_off = np.concatenate([[0], np.cumsum(want)]).tolist()
    for rank, lens, offs in res:
        assert lens == want, rank
        assert offs == want_off, rank


def test_c_abi_sharding_helpers_agree_with_the_python_side():
    """sz_shard_range / sz_stream_offsets (include/snappy_hip.h, host arithmetic: no GPU): what a C, JNI or cgo host
    uses for the two steps sharding.py does in Python -- the same ranges, the same layout, the same refusals."""
    import ctypes
    from pkg_loader import load_package
    P = load_package()
    from compression_accelerator_amd import sharding
    L = P.lib()
    u64 = ctypes.c_uint64
    for n in (0, 1, 7, 8, 9, 65, 16386, 262144):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                b, e = u64(), u64()
                assert L.sz_shard_range(n, rank, world, ctypes.byref(b), ctypes.byref(e)) == 0
                assert (b.value, e.value) == sharding.shard_range(n, rank, world)
    b, e = u64(), u64()
    assert L.sz_shard_range(10, 2, 2, ctypes.byref(b), ctypes.byref(e)) == P.SZ_EINVAL
    assert L.sz_shard_range(10, 0, 0, ctypes.byref(b), ctypes.byref(e)) == P.SZ_EINVAL
    rs = np.random.RandomState(5)
    lens = rs.randint(1, 76490, size=1000).astype(np.uint32)
    offs = np.zeros(1001, dtype=np.uint64)
    bad = u64(12345)
    assert L.sz_stream_offsets(lens.ctypes.data, 1000, 5, 76544, offs.ctypes.data, ctypes.byref(bad)) == 0
    assert offs[0] == 5 and (np.diff(offs.astype(np.int64)) == lens).all()
    for marker in (0xffffffff, 0xfffffffe, 0xfffffffd, 76545):
        l2 = lens.copy()
        l2[321] = marker
        assert L.sz_stream_offsets(l2.ctypes.data, 1000, 5, 76544, offs.ctypes.data, ctypes.byref(bad)) == P.SZ_ECORRUPT
        assert bad.value == 321
    assert L.sz_stream_offsets(None, 0, 3, 76544, offs.ctypes.data, None) == 0 and offs[0] == 3


def test_shard_range_partitions_exactly():
    load_package()
    from compression_accelerator_amd import sharding
    for n in (0, 1, 7, 8, 48, 262144, 262147):
        for world in (1, 2, 3, 4, 8):
            r = [sharding.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_gather_is_identity():
    load_package()
    from compression_accelerator_amd import sharding
    lens = torch.tensor([5, 7, 11], dtype=torch.int32)
    a, o = sharding.gather_lengths(lens)
    assert a.tolist() == [5, 7, 11] and o.tolist() == [0, 5, 12, 23]
    a, o = sharding.gather_lengths(lens, check=True)
    assert o.tolist() == [0, 5, 12, 23]


def test_error_markers_in_lengths_are_reported_not_summed():
    """The kernels flag a rejected / poisoned block with 0xffffffff / 0xfffffffe in out_len, which the
    int32 tensors of the binding show as -1 / -2: offsets computed over them would be silently wrong."""
    load_package()
    from compression_accelerator_amd import sharding
    lens = torch.tensor([5, -1, 11], dtype=torch.int32)
    with pytest.raises(ValueError):
        sharding.gather_lengths(lens, check=True)
    with pytest.raises(ValueError):
        sharding.check_lengths(torch.tensor([5, 7, 80000], dtype=torch.int32), slot_stride=76544)
    sharding.check_lengths(torch.tensor([5, 7, 76544], dtype=torch.int32), slot_stride=76544)


def test_bench_launcher_starts_the_ranks_itself():
    """bench.py --gpus N without torchrun: bench.launch_ranks starts N ranks as a child process
    (python -m torch.distributed.run ...), relays rank 0's JSON line and the exit code.  Rehearsed
    here with two gloo ranks whose lengths come from the oracle (tests/bench_rank_stub.py)."""
    import json
    import bench
    stub = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_rank_stub.py")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    rc, line, out, err = bench.launch_ranks(2, stub, ["10"], env=env, timeout=300)
    assert rc == 0, err[-2000:]
    assert line is not None, out[-2000:]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["rccl_ranks_seen"] == 2 and r["gathered_layout_ok"] is True
    want = [len(oracle_lib.raw_compress(x)) for x in _blocks(10)]
    assert r["lens"] == want and r["total"] == sum(want)
    # a failing rank is reported through the exit code, not swallowed
    rc2, line2, _, _ = bench.launch_ranks(2, stub, ["not-a-number"], env=env, timeout=300)
    assert rc2 != 0 and line2 is None


def test_eight_ranks_ragged_total_and_a_rank_whose_lengths_do_not_arrive():
    """The first N = 8 run on hardware must verify itself: world_size 8 over gloo, a total that
    does not divide (43 blocks: shards of 6, 6, 6, 5, 5, 5, 5, 5), `rccl_ranks_seen` counted from per-rank tags that
    every rank publishes and every rank checks against the slice it received, the per-rank slices and the layout
    against the oracle.  Then the same with one rank's lengths zeroed on the way: seen == 7, layout not ok."""
    import json
    import bench
    load_package()
    from compression_accelerator_amd import sharding
    stub = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_rank_stub.py")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["OMP_NUM_THREADS"] = "1"
    rc, line, out, err = bench.launch_ranks(8, stub, ["43"], env=env, timeout=600)
    assert rc == 0 and line is not None, (out[-2000:], err[-2000:])
    r = json.loads(line)
    assert r["n_gpus"] == 8 and r["rccl_ranks_seen"] == 8 and r["gathered_layout_ok"] is True
    want = [len(oracle_lib.raw_compress(x)) for x in _blocks(43)]
    assert r["lens"] == want and r["total"] == sum(want)
    sizes = [e - b for b, e in (sharding.shard_range(43, k, 8) for k in range(8))]
    assert sizes == [6, 6, 6, 5, 5, 5, 5, 5]
    env["SZ_STUB_DROP_RANK"] = "5"
    rc, line, out, err = bench.launch_ranks(8, stub, ["43"], env=env, timeout=600)
    assert rc == 0 and line is not None, (out[-2000:], err[-2000:])
    r = json.loads(line)
    assert r["rccl_ranks_seen"] == 7 and r["gathered_layout_ok"] is False


def test_bench_arguments_default_to_one_gpu_and_short_runs():
    import bench
    a = bench.parse_args([])
    assert (a.gpus, a.steps, a.warmup, a.workload, a.blocks_per_gpu) == (1, 20, 3, "c3", 32768)
