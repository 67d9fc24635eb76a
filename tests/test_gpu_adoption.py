"""The end of a launch (round 6): a team of the team kernel that finds the queue empty ADOPTS the block a lone wave of its
own workgroup is inside -- the lone wave hands its parse state over where a tile ends, the team copies its hash table
out of memory and goes on.  The bytes must be the oracle's whoever finished the block, a length has one writer, and the
path must really be taken: a diagnostic build (`-DSZ_EXP_COUNT`) marks the length of an adopted block with bit 29
(lone waves bit 31, the single kernel bit 30).  `SZ_NO_ADOPT=1` ends a launch as round 5 did (A/B runs): no marks."""
import ctypes
import os
import sys

import numpy as np
import pytest

import build_util
import fuzzgen
import oracle_lib
import parity_util
from test_gpu_team_giveup import _build_variant, SLOT

pytestmark = pytest.mark.gpu

ADOPTED, SINGLE, LONE = 0x20000000, 0x40000000, 0x80000000
LEN = 0x1FFFFFFF


def _bind(path):
    lib = ctypes.CDLL(path)
    vp, sz, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
    lib.sz_compress_blocks_hip.argtypes = [vp, sz, vp, u32, vp, sz, vp, u32, vp]
    lib.sz_compress_blocks_hip.restype = ctypes.c_int
    return lib


@pytest.fixture(scope="module")
def marked_lib(tmp_path_factory):
    return _bind(_build_variant(str(tmp_path_factory.mktemp("adopt")), ["-DSZ_EXP_COUNT"]))


def _launch(lib, src, nb, in_len=None, stride=65536):
    import torch
    dev = src.device
    dst = torch.full((nb * SLOT,), 0xA5, dtype=torch.uint8, device=dev)
    ol = torch.full((nb,), 0x5A5A5A5A, dtype=torch.int32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.sz_compress_blocks_hip(src.data_ptr(), stride, in_len.data_ptr() if in_len is not None else None, nb,
                                    dst.data_ptr(), SLOT, ol.data_ptr(), 1, stream)
    assert rc == 0
    torch.cuda.synchronize()
    return dst, ol.cpu().numpy().view(np.uint32)


@pytest.mark.timeout(400, method="thread")
def test_adopted_blocks_carry_the_oracles_bytes(marked_lib, monkeypatch):
    import torch
    monkeypatch.delenv("SZ_KERNEL", raising=False)
    monkeypatch.delenv("SZ_NO_ADOPT", raising=False)
    sys.path.insert(0, build_util.ROOT)
    import bench
    dev = torch.device("cuda:0")
    seen = {}
    # launches of text of several sizes (how a launch ends depends on how many blocks each worker gets), the C4 mix
    # (routed: the team kernel works through a queue of block numbers), the low-entropy workload (long matches: a lone
    # wave answers a request only between tiles)
    for kind, nb in (("text", 3072), ("text", 4096), ("text", 5000), ("text", 9001), ("mix", 6144), ("c5", 4096)):
        src = bench.make_input(kind, 0, nb, dev)
        for rep in range(2):  # (twice: the adoption words in LDS and the launch state are set up per launch)
            dst, marks = _launch(marked_lib, src, nb)
            plain = torch.from_numpy((marks & LEN).astype(np.int32)).to(dev)
            parity_util.assert_all_blocks_match_oracle(src, dst, plain, nb, SLOT, what="%s x %d, launch %d" % (kind, nb, rep))
            by_team_kernel = (marks & SINGLE) == 0
            adopted = (marks & ADOPTED) != 0
            assert not (adopted & ((marks & (LONE | SINGLE)) != 0)).any(), "an adopted block's length has one writer: the team"
            seen[(kind, nb)] = seen.get((kind, nb), 0) + int(adopted[by_team_kernel].sum())
    print("adopted blocks per launch shape (two launches each):", seen)
    # (3 072 blocks = one per worker: every team is through its block when its lone waves are half through theirs;
    # whether the other shapes end with something worth adopting is a matter of timing)
    assert seen[("text", 3072)] > 0 and sum(seen.values()) > seen[("text", 3072)], seen
    # ... and switched off, a launch ends as in round 5
    monkeypatch.setenv("SZ_NO_ADOPT", "1")
    src = bench.make_input("text", 0, 4096, dev)
    dst, marks = _launch(marked_lib, src, 4096)
    assert not ((marks & ADOPTED) != 0).any()
    plain = torch.from_numpy((marks & LEN).astype(np.int32)).to(dev)
    parity_util.assert_all_blocks_match_oracle(src, dst, plain, 4096, SLOT, what="text x 4096, no adoption")


@pytest.mark.timeout(400, method="thread")
def test_adoption_in_small_launches_of_every_kind_of_block(marked_lib, monkeypatch):
    """`SZ_KERNEL=team` from two blocks up: every workgroup's team runs out of blocks almost at once and asks its lone
    waves for theirs -- blocks of every generator, ragged lengths (only blocks of 16 KiB and more, with 4 KiB still
    to parse, are handed over: the others must be left alone), each against the oracle."""
    import torch
    monkeypatch.setenv("SZ_KERNEL", "team")
    monkeypatch.delenv("SZ_NO_ADOPT", raising=False)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(606)
    adopted_total = 0
    for trial in range(40):
        nb = int(rng.integers(2, 48))
        blocks = []
        for i in range(nb):
            n = int(rng.choice([65536, 65536, 65536, 40000, 20000, 16384, 16399, 12000, 4097, 900]))
            blocks.append(_one_block(rng, n))
        lens = np.array([len(b) for b in blocks], dtype=np.uint32)
        host = np.zeros(nb * 65536, dtype=np.uint8)
        for i, b in enumerate(blocks):
            host[i * 65536:i * 65536 + len(b)] = np.frombuffer(b, dtype=np.uint8)
        src = torch.from_numpy(host).to(dev)
        in_len = torch.from_numpy(lens.view(np.int32)).to(dev)
        dst, marks = _launch(marked_lib, src, nb, in_len=in_len)
        out = dst.cpu().numpy().reshape(nb, SLOT)
        for i, b in enumerate(blocks):
            want = oracle_lib.raw_compress(b)
            got_len = int(marks[i] & LEN)
            assert got_len == len(want), "trial %d block %d (n=%d): length %d, oracle %d (marks %#x)" % (trial, i, len(b), got_len, len(want), marks[i])
            assert out[i, :got_len].tobytes() == want, "trial %d block %d (n=%d): bytes differ (marks %#x)" % (trial, i, len(b), marks[i])
            if marks[i] & ADOPTED:
                assert len(b) >= 16384 + 15, "a block of %d bytes was handed over" % len(b)
                adopted_total += 1
    print("adopted in small team launches:", adopted_total)
    assert adopted_total > 0


def _one_block(rng, n):
    kind = int(rng.integers(0, 6))
    if kind == 0:
        text = fuzzgen.corpus("mtg")
        o = int(rng.integers(0, len(text) - n))
        return bytes(text[o:o + n])
    if kind == 1:  # low entropy: a short pattern repeated, with a few edits
        pat = rng.integers(0, 256, size=int(rng.integers(3, 700)), dtype=np.uint8)
        a = np.resize(pat, n).copy()
        for _ in range(int(rng.integers(0, 20))):
            a[int(rng.integers(0, n))] ^= 0x55
        return a.tobytes()
    if kind == 2:  # text with incompressible stretches (the fast dense path hands over to the general code and back)
        text = fuzzgen.corpus("mtg")
        o = int(rng.integers(0, len(text) - n))
        a = np.frombuffer(bytes(text[o:o + n]), dtype=np.uint8).copy()
        for _ in range(int(rng.integers(1, 6))):
            p = int(rng.integers(0, n))
            ln = int(rng.integers(40, 3000))
            a[p:p + ln] = rng.integers(0, 256, size=len(a[p:p + ln]), dtype=np.uint8)
        return a.tobytes()
    if kind == 3:  # few symbols: many bucket collisions inside a tile
        return rng.integers(0, 4, size=n, dtype=np.uint8).tobytes()
    if kind == 4:  # incompressible
        return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    # words of a small vocabulary
    vocab = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 9)), dtype=np.uint8)) + b" " for _ in range(int(rng.integers(5, 200)))]
    out = bytearray()
    while len(out) < n:
        out += vocab[int(rng.integers(0, len(vocab)))]
    return bytes(out[:n])
