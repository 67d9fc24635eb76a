"""-m gpu: the multi-GPU leg on hardware, inside the driver's own GPU run.

The one collective of the path -- the RCCL all_gather of the per-block lengths + prefix sum -- runs in a
CHILD process started by bench.launch_ranks (a fresh python under torch.distributed.run; this pytest
process, which has initialised the GPU, is never replaced).  With one GPU the group has one rank (the
`nccl` backend, communicator and stream choreography are the real ones); with two or more GPUs a twin
test runs two ranks and calls the C ABI on device index 1."""
import ctypes
import json
import os

import numpy as np
import pytest

import bench
import oracle_lib
from gpu_util import pkg

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run_bench(n_ranks, extra):
    args = ["--gpus", str(n_ranks), "--steps", "2", "--warmup", "1", "--blocks-per-gpu", "4096",
            "--no-cpu", "--no-extra"] + extra
    rc, line, out, err = bench.launch_ranks(n_ranks, BENCH, args, timeout=900)
    assert rc == 0, "bench ranks failed (rc %d)\n%s\n%s" % (rc, out[-2000:], err[-4000:])
    assert line is not None, "no JSON line from rank 0\n%s\n%s" % (out[-2000:], err[-2000:])
    return json.loads(line)


def test_rccl_length_gather_runs_on_hardware_with_one_rank():
    """bench.py --force-dist as a child process: init_process_group("nccl"), the all_gather of the lengths on
    a side stream under the next step's kernel, the layout every rank would pack with."""
    r = _run_bench(1, ["--force-dist"])
    assert r["n_gpus"] == 1
    assert r["rccl_ranks_seen"] == 1
    assert r["gathered_layout_ok"] is True
    assert r["blocks_differing_from_oracle"] == 0 and r["bit_exact_vs_oracle_all_blocks"] is True
    assert r["scaling"] == "weak" and r["config"]["blocks_per_gpu"] == 4096


def _device_count():
    import torch
    return torch.cuda.device_count()


def _run_c_ranks(tmp_path, world, data):
    """drivers/shard_compress.c, `world` processes (one per GPU): the path's sequence in plain C with librccl."""
    import subprocess
    import build_util
    exe = build_util.cc_rccl(os.path.join(build_util.DRIVERS, "shard_compress.c"), str(tmp_path / "shard_compress"))
    inp = tmp_path / "job.bin"
    inp.write_bytes(data)
    idf = str(tmp_path / "rccl.id")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([exe, str(inp), str(world), str(r), idf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
             for r in range(world)]
    outs = []
    try:
        for r, p in enumerate(procs):
            out, err = p.communicate(timeout=300)
            assert p.returncode == 0, "rank %d of the C caller failed (rc %d)\n%s" % (r, p.returncode, err.decode()[-3000:])
            outs.append(json.loads(out.decode().strip().splitlines()[-1]))
    finally:  # whatever happened -- a rank that failed, a timeout --: no rank is left behind holding a GPU
        for q in procs:
            if q.poll() is None:
                q.kill()
                q.wait()
    return outs


def _check_c_ranks(outs, world, data):
    n_blocks = (len(data) + 65535) // 65536
    want = [len(oracle_lib.compress_fragment(data[i * 65536:(i + 1) * 65536])) for i in range(n_blocks)]  # bodies (flags = 0)
    hdr = len(oracle_lib.raw_compress(data)) - sum(want)  # the stream's preamble
    for r, o in enumerate(outs):
        assert (o["rank"], o["world"], o["n_blocks"]) == (r, world, n_blocks)
        assert o["rccl_ranks_seen"] == world and o["own_slice_ok"] and o["layout_ok"]
        assert o["lens"] == want, "rank %d: gathered lengths differ from the oracle's" % r
        assert o["stream_bytes"] == hdr + sum(want) == len(oracle_lib.raw_compress(data))
    assert [o["begin"] for o in outs] == [outs[k - 1]["end"] if k else 0 for k in range(world)] and outs[-1]["end"] == n_blocks


def test_c_caller_with_rccl_directly_one_rank(tmp_path):
    """hipSetDevice, sz_shard_range, sz_compress_blocks_hip, ncclAllGather FROM LIBRCCL, sz_stream_offsets
    -- the sequence of INTEGRATION.md section 6 compiled and run, one rank; lengths and layout against the oracle."""
    import fuzzgen
    data = fuzzgen.corpus("mtg")[:65536 * 37 + 1234]  # a ragged job
    _check_c_ranks(_run_c_ranks(tmp_path, 1, data), 1, data)


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs")
def test_c_caller_with_rccl_directly_two_ranks(tmp_path):
    import fuzzgen
    data = fuzzgen.corpus("mtg")[:65536 * 37 + 1234]  # 38 blocks: 19 + 19; the second shard ends ragged
    _check_c_ranks(_run_c_ranks(tmp_path, 2, data), 2, data)


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs")
def test_two_ranks_over_rccl_and_the_c_abi_on_device_one():
    r = _run_bench(2, [])
    assert r["n_gpus"] == 2
    assert r["rccl_ranks_seen"] == 2
    assert r["gathered_layout_ok"] is True
    assert r["blocks_differing_from_oracle"] == 0
    # the C ABI follows the caller's current device: the same call on device index 1
    import torch
    import fuzzgen
    P = pkg()
    with torch.cuda.device(1):
        dev = torch.device("cuda", 1)
        data = fuzzgen.corpus("mtg")[:8 * 65536]
        src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev)
        dst, out_len = P.compress_blocks(src, 65536, 8)
        torch.cuda.synchronize(dev)
        out = dst.cpu().numpy().reshape(8, P.SLOT_STRIDE)
        ol = out_len.cpu().numpy()
        for i in range(8):
            assert out[i, :ol[i]].tobytes() == oracle_lib.raw_compress(data[i * 65536:(i + 1) * 65536])
