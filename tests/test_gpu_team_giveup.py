"""The team kernel's time-out paths: a build whose waves give up after TWO polls (`-DSZ_TEAM_SPIN_CAP=2u`)
instead of after seconds.  Whatever the waves of a team do then -- lose a barrier, lose a token, leave the persistent
loop -- the launch must end, every block's length must have been written exactly once with either the oracle's
length (and then the oracle's bytes) or the poison marker 0xfffffffe, and never anything else.  The lone waves of
the workgroups (and the teams that lose nothing) still finish blocks, so both outcomes occur."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

import fuzzgen
import oracle_lib
from build_util import CSRC, INC

pytestmark = pytest.mark.gpu

SOURCES = ["snappy_compress.hip", "snappy_pack.hip", "snappy_uncompress.hip", "snappy_stream_index.hip",
           "snappy_hwcompat.hip", "snappy_abi.cpp"]
SLOT = 76544
POISON = 0xFFFFFFFE


def _build_variant(tmp, defines):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INC, "-I" + CSRC] + defines
    procs = []
    for s in SOURCES:
        o = os.path.join(tmp, s.rsplit(".", 1)[0] + ".o")
        procs.append((o, subprocess.Popen(base + ["-c", os.path.join(CSRC, s), "-o", o], stderr=subprocess.DEVNULL)))
    for o, p in procs:
        assert p.wait() == 0 and os.path.getsize(o) > 0, "compile failed: " + o
    lib = os.path.join(tmp, "libsnappy_hip_giveup.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [o for o, _ in procs])
    return lib


@pytest.mark.timeout(170, method="thread")
def test_a_team_that_gives_up_poisons_its_block_once_and_the_launch_ends(tmp_path, monkeypatch):
    import torch
    lib = ctypes.CDLL(_build_variant(str(tmp_path), ["-DSZ_TEAM_SPIN_CAP=2u"]))
    vp, sz, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
    lib.sz_compress_blocks_hip.argtypes = [vp, sz, vp, u32, vp, sz, vp, u32, vp]
    lib.sz_compress_blocks_hip.restype = ctypes.c_int
    monkeypatch.setenv("SZ_KERNEL", "team")
    text = fuzzgen.corpus("mtg")
    nb = 1536
    blocks = [bytes(text[(i * 7919) % (len(text) - 65536):][:65536]) for i in range(24)]
    host = np.frombuffer(b"".join(blocks[i % 24] for i in range(nb)), dtype=np.uint8).copy()
    src = torch.from_numpy(host).cuda()
    want = [oracle_lib.raw_compress(b) for b in blocks]
    seen_ok = seen_poison = 0
    for rep in range(2):
        dst = torch.full((nb * SLOT,), 0xA5, dtype=torch.uint8, device="cuda")
        ol = torch.full((nb,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        rc = lib.sz_compress_blocks_hip(src.data_ptr(), 65536, None, nb, dst.data_ptr(), SLOT, ol.data_ptr(), 1,
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()  # (the launch ends: no wave waits for ever on a team mate that left)
        lens = ol.cpu().numpy().view(np.uint32)
        out = dst.cpu().numpy().reshape(nb, SLOT)
        for i in range(nb):
            if lens[i] == POISON:
                seen_poison += 1
                continue
            w = want[i % 24]
            assert lens[i] == len(w), "block %d: length %#x is neither the oracle's %d nor the poison marker" % (i, lens[i], len(w))
            assert out[i, :len(w)].tobytes() == w, "block %d reported as done but its bytes differ from the oracle" % i
            seen_ok += 1
    assert seen_ok > 0, "no block finished at all"
    assert seen_poison > 0, "the give-up path was never taken: the variant does not exercise what it is for"
