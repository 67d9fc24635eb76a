"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the oracle on the same seeded inputs and against the committed goldens (bit-exact)."""
import hashlib
import json
import os

import numpy as np
import pytest

import fuzzgen
import oracle_lib
from gpu_util import compress_many, pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "single", "team", "dual"], autouse=True)
def kernel_mode(request, monkeypatch):
    """Every parity test runs through every compress kernel: `single` (one wave per SIMD, LDS table),
    `team` (two waves share a block and its LDS table, parse token in LDS, + two global-table waves per
    workgroup: round 3's kernel for compressible launches) and `dual` (rounds 1-2's multi-wave kernel: three
    independent waves per SIMD), plus the default data-sampled choice.  The library reads SZ_KERNEL on every call."""
    if request.param == "auto":
        monkeypatch.delenv("SZ_KERNEL", raising=False)
    else:
        monkeypatch.setenv("SZ_KERNEL", request.param)
    return request.param

G = fuzzgen.GOLDEN_DIR


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def _first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return i
    return n


def _assert_same(got, want, what):
    assert got == want, "%s: GPU %d bytes vs oracle %d bytes, first difference at %d" % (
        what, len(got), len(want), _first_diff(got, want))


def test_native_library_is_loaded_and_sees_the_gpu():
    P = pkg()
    assert P.device_count() >= 1
    assert os.path.exists(P.LIB_PATH)
    with open("/proc/self/maps") as f:
        assert "libsnappy_hip.so" in f.read()
    # ... and it is THIS tree's library: the box loads whatever .so was pushed, so the identity of the sources it was
    # built from (csrc/Makefile SOURCE_ID, inside sz_version()) is recomputed from the sources beside it
    import build_util
    ver = P.lib().sz_version().decode()
    assert "src " + build_util.source_id() in ver, \
        "libsnappy_hip.so was not built from this tree's csrc/ + include/ (%s; sources: %s): run make" % (
            ver, build_util.source_id())
    print(ver)


def test_edge_sizes_all_kinds():
    sizes = [0, 1, 2, 3, 4, 5, 13, 14, 15, 16, 17, 18, 19, 20, 31, 32, 33, 59, 60, 61, 62, 63, 64, 65, 66, 67,
             68, 69, 127, 128, 129, 255, 256, 257, 258, 1000, 2047, 2048, 2049, 4096, 5000, 16383, 16384,
             16385, 65521, 65535, 65536]
    blocks, names = [], []
    seed = 9000
    for kind in fuzzgen.KINDS:
        for n in sizes:
            blocks.append(fuzzgen.gen(kind, seed, n).tobytes())
            names.append((kind, seed, n))
            seed += 1
    got = compress_many(blocks)
    for g, b, nm in zip(got, blocks, names):
        _assert_same(g, oracle_lib.raw_compress(b), str(nm))


def test_body_only_mode_has_no_preamble():
    blocks = [fuzzgen.gen(k, 7 + i, n).tobytes() for i, (k, n) in enumerate(
        [("text_mtg", 65536), ("random", 100), ("lz", 30000), ("same", 65536), ("words", 0)])]
    got = compress_many(blocks, varint=False)
    for g, b in zip(got, blocks):
        _assert_same(g, oracle_lib.compress_fragment(b), "body-only %d" % len(b))


def test_fuzz_goldens_from_libsnappy_1_1_8():
    with open(os.path.join(G, "golden_fuzz.json")) as f:
        cases = [c for c in json.load(f)["cases"] if c["n"] <= 65536]
    blocks = [fuzzgen.gen(c["kind"], c["seed"], c["n"]).tobytes() for c in cases]
    got = compress_many(blocks)
    bad = [(c["kind"], c["seed"], c["n"]) for c, g in zip(cases, got)
           if len(g) != c["out_len"] or sha(g) != c["out_sha256"]]
    assert not bad, bad[:10]


def test_file_goldens_per_block():
    with open(os.path.join(G, "golden_files.json")) as f:
        files = json.load(f)["files"]
    import test_oracle
    for e in files:
        data = test_oracle._input_for(e["name"])
        blocks = [data[o:o + 65536] for o in range(0, len(data), 65536)] or [b""]
        got = compress_many(blocks)
        if "blocks" in e:
            for k, (g, gb) in enumerate(zip(got, e["blocks"])):
                assert (len(g), sha(g)) == (gb["out_len"], gb["out_sha256"]), "%s block %d" % (e["name"], k)
        if len(blocks) == 1:
            assert (len(got[0]), sha(got[0])) == (e["out_len"], e["out_sha256"]), e["name"]
            if "out_hex" in e:
                assert got[0].hex() == e["out_hex"]


def test_random_differential_vs_oracle():
    rs = np.random.RandomState(4242)
    blocks = []
    for i in range(1500):
        kind = fuzzgen.KINDS[int(rs.randint(0, len(fuzzgen.KINDS)))]
        n = int(rs.randint(0, 3000)) if i % 4 else int(rs.randint(0, 65537))
        blocks.append(fuzzgen.gen(kind, 500000 + i, n).tobytes())
    got = compress_many(blocks)
    for i, (g, b) in enumerate(zip(got, blocks)):
        _assert_same(g, oracle_lib.raw_compress(b), "case %d" % i)


def test_bucket_collision_stress():
    """Inputs engineered so that many probes of one speculative batch share a hash bucket
    (the intra-batch ordering path of the kernel): few distinct 4-byte shingles, short periods."""
    rs = np.random.RandomState(99)
    blocks = []
    for period in (1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129):
        pat = rs.randint(0, 256, size=period).astype(np.uint8)
        for n in (200, 4096, 65536):
            a = np.resize(pat, n).copy()
            blocks.append(a.tobytes())
            # same thing with sparse random corruptions so matches keep breaking
            k = max(1, n // 97)
            a[rs.randint(0, n, size=k)] ^= rs.randint(1, 256, size=k).astype(np.uint8)
            blocks.append(a.tobytes())
    got = compress_many(blocks)
    for i, (g, b) in enumerate(zip(got, blocks)):
        _assert_same(g, oracle_lib.raw_compress(b), "collision case %d (n=%d)" % (i, len(b)))


def test_unaligned_buffers_and_odd_strides():
    """Nothing in the ABI promises alignment: source and destination at odd addresses, strides that
    are not multiples of anything, explicit per-block lengths including empty blocks."""
    import torch
    P = pkg()
    rs = np.random.RandomState(5)
    for (src_off, dst_off, in_stride, dst_stride) in ((1, 3, 65536, 76501), (7, 0, 40001, 46800), (13, 5, 65535, 76543)):
        nb = 40
        kinds = ("text_mtg", "random", "lz", "words", "same")
        lens = [int(rs.randint(0, in_stride + 1)) if i % 5 else (in_stride if i % 10 else 0) for i in range(nb)]
        blocks = [fuzzgen.gen(kinds[i % len(kinds)], 900 + i, lens[i]).tobytes() for i in range(nb)]
        host = np.zeros(src_off + nb * in_stride + 64, dtype=np.uint8)
        for i, b in enumerate(blocks):
            host[src_off + i * in_stride:src_off + i * in_stride + len(b)] = np.frombuffer(b, dtype=np.uint8)
        d_src = torch.from_numpy(host).cuda()
        d_len = torch.from_numpy(np.array(lens, dtype=np.int32)).cuda()
        d_dst = torch.full((dst_off + nb * dst_stride + 64,), 0xA5, dtype=torch.uint8, device="cuda")
        out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
        P.compress_blocks(d_src[src_off:], in_stride, nb, in_len=d_len, dst=d_dst[dst_off:], dst_stride=dst_stride,
                          out_len=out_len)
        torch.cuda.synchronize()
        ol = out_len.cpu().numpy()
        out = d_dst.cpu().numpy()
        assert (out[:dst_off] == 0xA5).all()
        for i, b in enumerate(blocks):
            base = dst_off + i * dst_stride
            want = oracle_lib.raw_compress(b)
            assert ol[i] == len(want) and out[base:base + ol[i]].tobytes() == want, (src_off, i, lens[i])
            assert (out[base + ol[i]:base + dst_stride] == 0xA5).all(), "wrote past the end of block %d" % i


def test_fixed_size_small_blocks_have_their_own_kernel():
    """Fixed blocks of <= 8 KiB (pages) go through compress_small_blocks_kernel (dynamic LDS table of
    2*T bytes), larger fixed blocks through the usual kernels: same bytes as the oracle for every
    table size 256 .. 16384 and every generator."""
    import torch
    P = pkg()
    for bs in (15, 16, 100, 256, 257, 1000, 1024, 4096, 5000, 8192, 12345, 16384):
        blocks = []
        for i, kind in enumerate(fuzzgen.KINDS * 2):
            blocks.append(fuzzgen.gen(kind, 7000 + 31 * bs + i, bs).tobytes())
        nb = len(blocks)
        src = torch.from_numpy(np.frombuffer(b"".join(blocks), dtype=np.uint8).copy()).cuda()
        stride = (int(P.max_compressed_length(bs)) + 8 + 15) // 16 * 16
        dst, ol = P.compress_blocks(src, bs, nb, dst_stride=stride)
        torch.cuda.synchronize()
        ln = ol.cpu().numpy()
        out = dst.reshape(nb, stride).cpu().numpy()
        for i in range(nb):
            _assert_same(out[i, :ln[i]].tobytes(), oracle_lib.raw_compress(blocks[i]), "bs=%d block %d" % (bs, i))


def test_ten_thousand_small_blocks_every_one_against_the_oracle():
    """>= 10 000 fixed small blocks (4 KiB and 1 KiB pages of text, and of the c3/text/c5 mix) in one
    launch each, every block compared with the oracle."""
    import sys
    import torch
    import build_util
    import parity_util
    sys.path.insert(0, build_util.ROOT)
    import bench
    P = pkg()
    dev = torch.device("cuda:0")
    for kind, bs, n64 in (("text", 4096, 1024), ("mix", 4096, 768), ("text", 1024, 192)):
        src = bench.make_input(kind, 0, n64, dev)  # n64 x 64 KiB, cut into pages
        nb = n64 * 65536 // bs
        assert nb >= 10000
        stride = (int(P.max_compressed_length(bs)) + 8 + 15) // 16 * 16
        dst, ol = P.compress_blocks(src, bs, nb, dst_stride=stride)
        torch.cuda.synchronize()
        parity_util.assert_all_blocks_match_oracle(src, dst, ol, nb, stride, what="%s pages of %d" % (kind, bs), block=bs)


def test_long_matches_of_every_length_class_offset_and_alignment():
    """The long match extension works in steps (first bytes up to a 16-byte boundary of the input,
    whole 4 KiB steps, a masked last step, the final < 16 bytes): blocks of random bytes with ONE
    planted repeat of length L at distance d, started at every alignment class and ended by a
    differing byte or by the block end, against the oracle."""
    rng = np.random.default_rng(20261004)
    blocks, names = [], []
    lengths = [33, 47, 48, 49, 63, 64, 65, 80, 81, 95, 96, 97, 111, 112, 113, 1023, 1024, 1025, 1100, 4095, 4096,
               4097, 4111, 4112, 4113, 4160, 4200, 8191, 8192, 8208, 8209, 12000, 20000]
    dists = [1, 2, 3, 5, 15, 16, 17, 64, 100, 4096, 4099, 20000]
    k = 0
    for L in lengths:
        for d in dists:
            start = 20001 + (k % 16)  # the copy starts at every residue mod 16 over the cases
            k += 1
            n = 65536
            b = rng.integers(0, 256, n, dtype=np.uint8)
            for i in range(L):  # (a repeat at distance d < L overlaps itself, as in a run)
                b[start + i] = b[start + i - d]
            if start + L < n:
                b[start + L] = b[start + L - d] ^ 0x5A
            blocks.append(b.tobytes())
            names.append(("stop", L, d, start))
    for d in dists:  # ... and repeats that run into the block end, for every length of the last partial step
        for tail in (0, 1, 15, 16, 17, 31, 100, 1023, 1024, 1025, 4095):
            n = 65536 - tail
            start = 30000 + (d % 7)
            b = rng.integers(0, 256, n, dtype=np.uint8)
            for i in range(start, n):
                b[i] = b[i - d]
            blocks.append(b.tobytes())
            names.append(("to-end", n, d, start))
    got = compress_many(blocks)
    for g, b, nm in zip(got, blocks, names):
        _assert_same(g, oracle_lib.raw_compress(b), str(nm))
