"""GPU tests of the drop-in boundary and of the rows next to the hot path (pack, decompress):
host-buffer API (snappy::Compress semantics), RoCC command surface, the reference-shaped C/C++
drivers, device-side stream packing, batch decompression, and full-size properties."""
import ctypes
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import build_util
import fuzzgen
import oracle_lib
import parity_util
from gpu_util import compress_many, pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "single", "pair"], autouse=True)
def decode_kernel(request, monkeypatch):
    """Every test of this file through both decode kernels (SZ_DECODE_KERNEL: one wave per stream / parser + mover)
    and through the library's own choice."""
    if request.param != "auto":
        monkeypatch.setenv("SZ_DECODE_KERNEL", request.param)
    return request.param
G = fuzzgen.GOLDEN_DIR


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


# ------------------------------------------------------------------ host-buffer API

@pytest.mark.parametrize("mode", ["auto", "team", "dual"])
def test_host_compress_is_snappy_compress(mode, monkeypatch):
    if mode != "auto":
        monkeypatch.setenv("SZ_KERNEL", mode)  # ragged last fragment through the multi-wave kernel too
    P = pkg()
    mtg = fuzzgen.corpus("mtg")
    cases = [b"", b"a", b"ab" * 7, mtg[:65535], mtg[:65536], mtg[:65537], mtg[:200001],
             fuzzgen.gen("lz", 3, 300000).tobytes(), fuzzgen.corpus("random_ascii")]
    for data in cases:
        got = P.compress(data)
        assert got == oracle_lib.raw_compress(data), len(data)
        assert P.uncompress(got) == data
    with open(os.path.join(G, "golden_files.json")) as f:
        gold = {e["name"]: e for e in json.load(f)["files"]}
    whole = P.compress(mtg)
    e = gold["all-mtg-cards.txt"]
    assert (len(whole), sha(whole)) == (e["out_len"], e["out_sha256"])
    assert "%016x" % oracle_lib.fnv1a64(whole) == e["out_fnv1a64"] == "35d49914bb85af93"
    assert P.uncompress(whole) == mtg



@pytest.mark.timeout(300, method="thread")
@pytest.mark.parametrize("register", ["registered", "pageable", "read_only_mapping", "copy_out_kernel", "copy_out_engine"])
def test_host_compress_pipeline_of_large_buffers(register, monkeypatch, tmp_path):
    """sz_compress on a buffer of 128 MiB or more runs as a pipeline of stages (copy in | compress + pack | copy out, three
    streams, the caller's memory registered for DMA for the length of the call): three stages with a ragged last one,
    text, incompressible and low-entropy data side by side, bytes == the oracle's snappy::Compress.  The same without
    registration (SZ_NO_HOST_REGISTER: the runtime's pageable copies), from a read-only file mapping (registration
    of PROT_READ pages fails or needs the read-only flag), with a `dst` that is too small (SZ_ENOSPC, and the next call
    is unharmed), and back through sz_uncompress.  A stage that did not compress is copied out by a kernel that stores
    into the registered `dst`, the others by the DMA engines (round 6): `SZ_COPY_OUT=kernel` / `engine` take one way for
    every stage (the kernel's stretches then begin at every alignment)."""
    import mmap
    monkeypatch.delenv("SZ_COPY_OUT", raising=False)
    if register == "pageable":
        monkeypatch.setenv("SZ_NO_HOST_REGISTER", "1")
    if register.startswith("copy_out_"):
        monkeypatch.setenv("SZ_COPY_OUT", register[len("copy_out_"):])
    P = pkg()
    L = P.lib()
    mtg, rnd = fuzzgen.corpus("mtg"), fuzzgen.corpus("random_ascii")
    n = (200 << 20) + 12345
    parts = [(mtg * 30)[777:777 + (70 << 20)], (rnd * 600)[5:5 + (50 << 20)], (mtg[:4096] * 16) * 256]
    parts.append((mtg * 30)[100001:100001 + n - sum(len(p) for p in parts)])
    data = b"".join(parts)
    assert len(data) == n
    want = oracle_lib.raw_compress(data)
    if register == "read_only_mapping":
        path = tmp_path / "in.bin"
        path.write_bytes(data)
        f = open(path, "rb")
        m = mmap.mmap(f.fileno(), 0, prot=mmap.PROT_READ)
        src_holder = np.frombuffer(m, dtype=np.uint8)
    else:
        src_holder = np.frombuffer(data, dtype=np.uint8)
    cap = P.max_compressed_length(n)
    dst = np.empty(cap + 64, dtype=np.uint8)
    dst[cap:] = 0xA5
    got = ctypes.c_size_t(0)
    rc = L.sz_compress(src_holder.ctypes.data, n, dst.ctypes.data, cap, ctypes.byref(got))
    assert rc == 0 and got.value == len(want)
    assert dst[:got.value].tobytes() == want, "pipelined sz_compress differs from the oracle"
    assert (dst[cap:] == 0xA5).all()
    # a destination that holds the first stage's output but not the whole stream
    small = np.empty(len(want) - 1000 + 64, dtype=np.uint8)
    small[-64:] = 0x5A
    rc = L.sz_compress(src_holder.ctypes.data, n, small.ctypes.data, len(want) - 1000, ctypes.byref(got))
    assert rc == P.SZ_ENOSPC
    assert (small[-64:] == 0x5A).all()
    # ... leaves nothing behind: the same call again with room, and the way back
    rc = L.sz_compress(src_holder.ctypes.data, n, dst.ctypes.data, cap, ctypes.byref(got))
    assert rc == 0 and dst[:got.value].tobytes() == want
    back = np.empty(n, dtype=np.uint8)
    rc = L.sz_uncompress(dst.ctypes.data, got.value, back.ctypes.data, n, ctypes.byref(got))
    assert rc == 0 and got.value == n and back.tobytes() == data


def test_compress_begin_finish_protocol():
    """sz_compress in two halves (what snappy::Compress(..., std::string*) does around its resize): one call in flight per
    host thread, finish without begin refused, an abandoned call leaves the thread free, and every size class --
    nothing, one fragment, several fragments -- gives the bytes of sz_compress."""
    P = pkg()
    L = P.lib()
    mtg = fuzzgen.corpus("mtg")
    got = ctypes.c_size_t(0)
    out = ctypes.create_string_buffer(P.max_compressed_length(400000))
    assert L.sz_compress_finish(out, len(out), ctypes.byref(got)) == P.SZ_EINVAL  # nothing begun
    for n in (0, 1, 130, 65536, 65537, 400000):
        data = mtg[:n]
        src = ctypes.create_string_buffer(data, max(1, n))
        assert L.sz_compress_begin(src, n) == 0
        assert L.sz_compress_begin(src, n) == P.SZ_EINVAL  # a call is in flight on this thread
        assert L.sz_compress(src, n, out, len(out), ctypes.byref(got)) == P.SZ_EINVAL
        assert L.sz_compress_finish(out, len(out), ctypes.byref(got)) == 0
        assert out.raw[:got.value] == oracle_lib.raw_compress(data), n
        assert L.sz_compress_finish(out, len(out), ctypes.byref(got)) == P.SZ_EINVAL  # finished already
        # abandoned, then a destination that is too small: the thread is free again each time
        assert L.sz_compress_begin(src, n) == 0
        assert L.sz_compress_finish(None, 0, None) == 0
        assert L.sz_compress_begin(src, n) == 0
        assert L.sz_compress_finish(out, 0, ctypes.byref(got)) == P.SZ_ENOSPC
        assert L.sz_compress(src, n, out, len(out), ctypes.byref(got)) == 0
        assert out.raw[:got.value] == oracle_lib.raw_compress(data)
    # a call begun and never finished is drained by sz_release_thread_resources
    src = ctypes.create_string_buffer(mtg[:300000], 300000)
    assert L.sz_compress_begin(src, 300000) == 0
    L.sz_release_thread_resources()
    assert L.sz_compress(src, 300000, out, len(out), ctypes.byref(got)) == 0
    assert out.raw[:got.value] == oracle_lib.raw_compress(mtg[:300000])


def test_host_api_one_fragment_size_sweep():
    """a single fragment goes through the small-block kernel (n <= 8 192, one block of in_stride = n) or the
    single kernel (larger; the slot is sized to its bound and the length / status words lie right behind it): every
    size class, the 8192/8193 switch, sizes that are no multiple of 16, text and noise, each against the oracle with a
    round trip through sz_uncompress; and a damaged one-fragment stream is refused."""
    P = pkg()
    mtg, noise = fuzzgen.corpus("mtg"), fuzzgen.corpus("random_ascii")
    for n in (15, 16, 63, 64, 130, 511, 512, 4096, 8191, 8192, 8193, 16384, 65535):
        for data in (mtg[1000:1000 + n], noise[7:7 + n]):
            got = P.compress(data)
            assert got == oracle_lib.raw_compress(data), (n, data[:8])
            assert P.uncompress(got) == data, n
            if n >= 63:
                bad = bytearray(got)
                bad[len(bad) // 2] ^= 0x5a  # most flips break an element; every one must be refused or decode to something else, never crash
                try:
                    assert P.uncompress(bytes(bad)) != data or bytes(bad) == got
                except P.SnappyHipError as e:
                    assert e.code in (P.SZ_ECORRUPT, P.SZ_ENOSPC)
    with pytest.raises(P.SnappyHipError):
        P.uncompress(oracle_lib.raw_compress(mtg[:130])[:-3])  # the RoCC driver's size, truncated


def test_profile_hook_is_one_shot_on_every_path():
    """events handed to sz_profile_next_launch belong to the NEXT compress call whatever becomes of it: a
    call that fails its argument check, or takes the accelerator-format kernel, records nothing and disarms the hook;
    the call after it must not record into them."""
    import torch
    P = pkg()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    hip.hipEventQuery.argtypes = [ctypes.c_void_p]
    hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
    ev = []
    for _ in range(4):
        e = ctypes.c_void_p()
        assert hip.hipEventCreate(ctypes.byref(e)) == 0
        ev.append(e.value)
    dev = torch.device("cuda:0")
    nb = 64
    src = torch.zeros(nb * 65536, dtype=torch.uint8, device=dev)
    dst = torch.empty(nb * P.SLOT_STRIDE, dtype=torch.uint8, device=dev)
    ol = torch.zeros(nb, dtype=torch.int32, device=dev)
    t = ctypes.c_float()
    # (1) armed, then a call that fails its argument check (stride too small): disarmed, nothing recorded
    P.profile_next_launch(*ev)
    rc = P.lib().sz_compress_blocks_hip(ctypes.c_void_p(src.data_ptr()), 65536, None, nb, ctypes.c_void_p(dst.data_ptr()), 100,
                                        ctypes.c_void_p(ol.data_ptr()), 1, None)
    assert rc == P.SZ_ENOSPC
    P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol)  # an unrelated launch: must not touch the events
    torch.cuda.synchronize()
    assert hip.hipEventElapsedTime(ctypes.byref(t), ev[0], ev[1]) != 0, "a later, unrelated launch recorded into events of a failed call"
    hip.hipGetLastError()
    # (2) armed, then the accelerator-format kernel: disarmed as well
    P.profile_next_launch(*ev)
    P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol, hw_compat=True)
    P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol)
    torch.cuda.synchronize()
    assert hip.hipEventElapsedTime(ctypes.byref(t), ev[0], ev[1]) != 0
    hip.hipGetLastError()
    # (3) armed, then a real launch: the single kernel's pair is recorded; a launch of small fixed blocks records it too
    P.profile_next_launch(*ev)
    P.compress_blocks(src, 65536, nb, dst=dst, out_len=ol)
    torch.cuda.synchronize()
    assert hip.hipEventElapsedTime(ctypes.byref(t), ev[0], ev[1]) == 0 and t.value > 0
    P.profile_next_launch(*ev)
    P.compress_blocks(src, 4096, nb, dst=dst, out_len=ol)
    torch.cuda.synchronize()
    assert hip.hipEventElapsedTime(ctypes.byref(t), ev[0], ev[1]) == 0 and t.value > 0
    for e in ev:
        hip.hipEventDestroy(ctypes.c_void_p(e))


def test_host_uncompress_rejects_malformed():
    P = pkg()
    good = oracle_lib.raw_compress(b"hello hello hello hello hello hello, said the fox to the socks")
    assert P.uncompress(good).startswith(b"hello hello")
    bad_streams = [good[:-1], good[:3], good[:1] + b"\x05\x00\x10", good + b"\x00",
                   bytes([10, 0x09, 0x05, 0x00]),          # copy before any output (offset > produced)
                   bytes([4, 0x01 | (0 << 2), 0x00])]       # offset 0
    for bad in bad_streams:
        with pytest.raises(P.SnappyHipError) as e:
            P.uncompress(bad)
        assert e.value.code in (P.SZ_ECORRUPT, P.SZ_ENOSPC)


def test_rocc_command_surface_via_ctypes():
    P = pkg()
    data = fuzzgen.corpus("fox")
    src = ctypes.create_string_buffer(data, len(data))
    dst = ctypes.create_string_buffer(P.max_compressed_length(len(data)))
    back = ctypes.create_string_buffer(len(data))
    assert P.rocc_cmd(P.ROCC_SET_LENGTH, len(data)) == 0
    clen = P.rocc_cmd(P.ROCC_COMPRESS, ctypes.addressof(src), ctypes.addressof(dst))
    assert dst.raw[:clen] == oracle_lib.raw_compress(data) and clen == 2407
    assert P.rocc_cmd(P.ROCC_SET_LENGTH, clen) == 0
    assert P.rocc_cmd(P.ROCC_UNCOMPRESS, ctypes.addressof(dst), ctypes.addressof(back)) == 0
    assert back.raw == data
    # corrupt stream -> rd != 0, like the reference's "non-zero on failure" (tests/compression.c:30)
    assert P.rocc_cmd(P.ROCC_SET_LENGTH, clen - 1) == 0
    assert P.rocc_cmd(P.ROCC_UNCOMPRESS, ctypes.addressof(dst), ctypes.addressof(back)) != 0


# ------------------------------------------------------------------ reference-shaped drivers

def test_rocc_driver_program(tmp_path):
    exe = build_util.cc(os.path.join(build_util.DRIVERS, "rocc_compress.c"), str(tmp_path / "rocc_compress"))
    r = subprocess.run([exe], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == [b"130", b"112", b"0"]  # thistle sentence: SURVEY.md App. B
    fox = os.path.join(fuzzgen.DATA_DIR, "fox-in-socks.txt")
    out = str(tmp_path / "fox.snappy")
    r = subprocess.run([exe, fox, out], capture_output=True, timeout=120)
    assert r.returncode == 0 and r.stdout.split() == [b"4602", b"2407", b"0"]
    assert open(out, "rb").read() == oracle_lib.raw_compress(fuzzgen.corpus("fox"))


def test_jni_shim_executes_on_the_gpu(tmp_path):
    """The C side of the north star's Scala/JVM driver (drivers/jni/snappy_hip_jni.c), EXECUTED: built against the
    stand-in tests/jni_stub/jni.h and called by tests/jni_stub/jni_driver.c the way a JVM calls it -- direct buffers in,
    the RoCC command pair (SET_LENGTH, COMPRESS -> rd; SET_LENGTH, UNCOMPRESS -> rd) of tests/compression.c:20-39 inside.
    thistle 130 -> 112, fox == the oracle's bytes, a ragged multi-fragment text, round trips, a damaged stream
    (rd != 0), a dst one byte short (SZ_ENOSPC), and the device-pointer entry on blocks resident in HBM."""
    exe = build_util.cc_jni(str(tmp_path / "jni_driver"))
    thistle = (b"If Tid the thistle sifter lifts his sieve of sifted thistles, then he'll shift "
               b"the thistles in his sifting sieve to live unsifted.")
    inputs = [fuzzgen.corpus("mtg")[1000:1000 + 65536 * 3 + 1234], thistle, fuzzgen.corpus("fox"),
              fuzzgen.tiled_block(fuzzgen.corpus("random_ascii"), 1)]
    paths = []
    for k, b in enumerate(inputs):
        paths.append(str(tmp_path / ("in%d.bin" % k)))
        open(paths[-1], "wb").write(b)
    r = subprocess.run([exe, "gpu", str(tmp_path)] + paths, capture_output=True, timeout=170)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads(r.stdout.decode())
    for key, want in build_util.JNI_ARGUMENT_CASES.items():
        assert out[key] == want, key
    for k, (b, f) in enumerate(zip(inputs, out["files"])):
        want = oracle_lib.raw_compress(b)
        assert f["n"] == len(b) and f["compressed"] == len(want), (k, f)
        assert open(str(tmp_path / ("%d.snappy" % k)), "rb").read() == want, "file %d: compress() through the JNI shim differs from the oracle" % k
        assert f["uncompress_rd"] == 0 and f["round_trip"] is True
        assert f["damaged_rd"] not in (0, -1), f  # rd != 0 on failure (tests/compression.c:30), or SZ_ECORRUPT from the probe
        assert f["tight_dst"] == -28
    assert out["files"][1]["compressed"] == 112 and out["files"][2]["compressed"] == 2407
    dev = out["device"]
    assert dev["rc"] == 0 and dev["rc_tight_stride"] == -28 and dev["blocks"] == 3
    for b in range(3):
        want = oracle_lib.raw_compress(inputs[0][b * 65536:(b + 1) * 65536])
        assert dev["lens"][b] == len(want)
        assert open(str(tmp_path / ("dev_%d.snappy" % b)), "rb").read() == want


def test_sw_benchmark_driver_reproduces_reference_sizes(tmp_path):
    exe = build_util.cxx(os.path.join(build_util.DRIVERS, "sw_benchmark.cc"), str(tmp_path / "sw_benchmark"))
    with open(os.path.join(G, "sw_results_sizes.json")) as f:
        pins = json.load(f)
    known_delta = {("real", 10000), ("real", 20000), ("real", 50000)}
    exact = 0
    for typ, prefix in pins["file_prefix"].items():
        rows = [r for r in pins["rows"] if r["type"] == typ]
        files = [os.path.join(fuzzgen.DATA_DIR, "benchmark-data", "%s-%d.txt" % (prefix, r["length"])) for r in rows]
        r = subprocess.run([exe, typ] + files, capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.decode().split()
        assert lines[0] == "type;length;ns;compressed-length"
        for row, path, line in zip(rows, files, lines[1:]):
            t, length, _ns, clen = line.split(";")
            assert t == typ and int(length) == row["length"]
            assert int(clen) == len(oracle_lib.raw_compress(open(path, "rb").read()))
            if (typ, row["length"]) not in known_delta:
                assert int(clen) == row["compressed_length"]
                exact += 1
    assert exact == 33


# ------------------------------------------------------------------ pack + batch decompress

def test_device_pack_yields_the_whole_stream():
    import torch
    P = pkg()
    data = fuzzgen.corpus("mtg")[:65536 * 9 + 777]
    n = len(data)
    src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    slots, lens, nb = P.compress_stream(src, n)
    hdr = bytes([0x80 | (n & 0x7f), 0x80 | ((n >> 7) & 0x7f), (n >> 14) & 0x7f])
    packed, offs = P.pack_stream(slots, P.SLOT_STRIDE, lens, nb, base=len(hdr))
    torch.cuda.synchronize()
    total = int(offs[nb].item())
    got = hdr + packed[len(hdr):total].cpu().numpy().tobytes()
    assert got == oracle_lib.raw_compress(data)
    o = offs.cpu().numpy()
    l = lens.cpu().numpy()
    assert o[0] == len(hdr) and (np.diff(o) == l[:nb]).all()


def test_packed_stream_decodes_in_parallel_with_its_offsets():
    """sz_uncompress_packed_hip: the inverse of the pack step.  (1) what this build packed, (2) a
    stream that libsnappy 1.1.8 produced (golden), with fragment offsets recomputed by the oracle,
    (3) wrong offsets or sizes are reported per fragment, not followed."""
    import torch
    P = pkg()
    # (1) own stream, ragged last fragment
    for data in (fuzzgen.corpus("mtg")[:65536 * 9 + 777], fuzzgen.gen("lz", 8, 65536 * 3).tobytes(),
                 fuzzgen.corpus("random_ascii") + fuzzgen.corpus("random_ascii")[:31073]):
        n = len(data)
        src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        slots, lens, nb = P.compress_stream(src, n)
        nh = 1 if n < 128 else (2 if n < 16384 else (3 if n < 2097152 else 4))
        packed, offs = P.pack_stream(slots, P.SLOT_STRIDE, lens, nb, base=nh)
        back, ol, st = P.uncompress_packed(packed, offs, nb, n)
        torch.cuda.synchronize()
        assert int(st.abs().sum().item()) == 0
        assert back[:n].cpu().numpy().tobytes() == data
        want_len = [min(65536, n - 65536 * b) for b in range(nb)]
        assert ol.cpu().numpy().tolist() == want_len
    # (2) the golden whole-file stream of all-mtg-cards.txt (libsnappy 1.1.8): 48 fragments
    mtg = fuzzgen.corpus("mtg")
    whole = oracle_lib.raw_compress(mtg)
    with open(os.path.join(G, "golden_files.json")) as f:
        gold = {e["name"]: e for e in json.load(f)["files"]}
    assert sha(whole) == gold["all-mtg-cards.txt"]["out_sha256"]
    nb = (len(mtg) + 65535) // 65536
    body = [len(oracle_lib.raw_compress(mtg[b * 65536:(b + 1) * 65536])) - (3 if len(mtg[b * 65536:(b + 1) * 65536]) >= 16384 else 2)
            for b in range(nb)]
    nh = len(whole) - sum(body)
    offs = np.concatenate([[nh], nh + np.cumsum(body)]).astype(np.uint64)
    d_whole = torch.from_numpy(np.frombuffer(whole, dtype=np.uint8).copy()).cuda()
    d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
    back, ol, st = P.uncompress_packed(d_whole, d_offs, nb, len(mtg))
    torch.cuda.synchronize()
    assert int(st.abs().sum().item()) == 0
    assert back[:len(mtg)].cpu().numpy().tobytes() == mtg
    # (3) damaged side information
    bad = offs.copy()
    bad[5] += 1  # fragments 4 and 5 now start / end in the middle of an element
    back, ol, st = P.uncompress_packed(d_whole, torch.from_numpy(bad.view(np.int64)).cuda(), nb, len(mtg))
    torch.cuda.synchronize()
    stn = st.cpu().numpy()
    assert stn[4] != 0 or stn[5] != 0
    assert (np.delete(stn, [4, 5]) == 0).all()
    back, ol, st = P.uncompress_packed(d_whole, d_offs, nb, len(mtg) - 5)
    torch.cuda.synchronize()
    assert st.cpu().numpy()[-1] != 0  # the last fragment holds 5 bytes more than announced
    with pytest.raises(P.SnappyHipError):
        P.uncompress_packed(d_whole, d_offs, nb + 1, len(mtg))  # n_blocks must match total_ulen


def test_batch_decompress_round_trip_all_kinds():
    import torch
    P = pkg()
    rs = np.random.RandomState(11)
    blocks = []
    for i in range(400):
        kind = fuzzgen.KINDS[i % len(fuzzgen.KINDS)]
        n = int(rs.randint(0, 65537)) if i % 3 else int(rs.randint(0, 300))
        blocks.append(fuzzgen.gen(kind, 777000 + i, n).tobytes())
    comp = compress_many(blocks)
    nb = len(blocks)
    host = np.zeros((nb, P.SLOT_STRIDE), dtype=np.uint8)
    clen = np.zeros(nb, dtype=np.int32)
    for i, c in enumerate(comp):
        host[i, :len(c)] = np.frombuffer(c, dtype=np.uint8)
        clen[i] = len(c)
    d_src = torch.from_numpy(host).cuda().reshape(-1)
    d_len = torch.from_numpy(clen).cuda()
    dst, out_len, status = P.uncompress_blocks(d_src, P.SLOT_STRIDE, d_len, nb)
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == 0).all()
    ol = out_len.cpu().numpy()
    out = dst.cpu().numpy().reshape(nb, 65536)
    for i, b in enumerate(blocks):
        assert ol[i] == len(b) and out[i, :ol[i]].tobytes() == b, i


# ------------------------------------------------------------------ full-size properties

def _full_size_round_trip(kind, n_blocks, expect_len=None):
    """compress -> (GPU) decompress == input for every block; lengths sane; every block == oracle."""
    import sys
    import torch
    sys.path.insert(0, build_util.ROOT)
    import bench
    P = pkg()
    dev = torch.device("cuda:0")
    src = bench.make_input(kind, 0, n_blocks, dev)
    dst, out_len = P.compress_blocks(src, 65536, n_blocks)
    back, ulen, status = P.uncompress_blocks(dst, P.SLOT_STRIDE, out_len, n_blocks)
    torch.cuda.synchronize()
    assert int(status.abs().sum().item()) == 0
    assert bool((ulen == 65536).all().item())
    assert torch.equal(back, src)
    lens = out_len.cpu().numpy()
    assert lens.max() <= P.max_compressed_length(65536) + 3
    if expect_len is not None:
        # a few windows of random text do hold a chance 4-byte repeat that the probe schedule
        # hits (expected ~5e-4 per block)
        odd = np.nonzero(lens != expect_len)[0]
        assert len(odd) <= max(4, n_blocks // 500), len(odd)
    # EVERY block against the oracle, bytes and lengths (the round trip above only proves "valid Snappy")
    parity_util.assert_all_blocks_match_oracle(src, dst, out_len, n_blocks, P.SLOT_STRIDE, what="%s x %d" % (kind, n_blocks))
    return lens


@pytest.mark.parametrize("mode", ["single", "team", "dual"])
def test_full_size_forced_kernels(mode, monkeypatch):
    """Every compress kernel at full size on every workload (auto mode picks one per launch)."""
    monkeypatch.setenv("SZ_KERNEL", mode)
    _full_size_round_trip("c3", 4096, expect_len=65542)
    _full_size_round_trip("text", 2048)
    _full_size_round_trip("mix", 3072)


def test_a_compressible_minority_inside_an_incompressible_launch(monkeypatch):
    """4 096 incompressible blocks with every 16th replaced by text or a low-entropy block (and two empty-ish
    ones), default kernel choice: the launch sample meets more than one kind of block, so every block is routed by
    its own class -- the single kernel keeps the incompressible ones, the team kernel takes the rest
    (tests/test_gpu_routing.py checks who wrote what); every block equals the oracle and decodes."""
    import sys
    import torch
    sys.path.insert(0, build_util.ROOT)
    import bench
    monkeypatch.delenv("SZ_KERNEL", raising=False)
    P = pkg()
    dev = torch.device("cuda:0")
    nb = 4096
    src = bench.make_input("c3", 0, nb, dev).reshape(nb, 65536).clone()
    text = bench.make_input("text", 5, nb // 16, dev).reshape(-1, 65536)
    c5 = bench.make_input("c5", 9, nb // 32, dev).reshape(-1, 65536)
    src[0::16] = text
    src[8::32] = c5
    src[3] = 0            # one long run
    src[5, 100:] = 7
    src = src.reshape(-1).contiguous()
    dst, out_len = P.compress_blocks(src, 65536, nb)
    back, ulen, status = P.uncompress_blocks(dst, P.SLOT_STRIDE, out_len, nb)
    torch.cuda.synchronize()
    assert int(status.abs().sum().item()) == 0 and torch.equal(back, src)
    lens = out_len.cpu().numpy()
    assert (lens[1::16] == 65542).all() and (lens[0::16] < 40000).all()
    parity_util.assert_all_blocks_match_oracle(src, dst, out_len, nb, P.SLOT_STRIDE, what="compressible minority")


@pytest.mark.parametrize("mode", ["dual", "team"])
def test_global_tables_are_clean_for_every_block(mode, monkeypatch):
    """The multi-wave kernel's extra waves (and the team kernel's lone waves) keep their tables in global memory and zero them per block;
    the team kernel's teams reuse one LDS table, two tokens and a barrier counter block after block:
    nothing of one block (or one launch) may leak into the next.  Same two blocks, several hundred
    launches through the same table slots; then blocks of different sizes (different table sizes)
    alternating through one slot; then the launch that hung experimental builds (incompressible blocks
    only)."""
    import torch
    monkeypatch.setenv("SZ_KERNEL", mode)
    P = pkg()
    blocks = [fuzzgen.gen("text_mtg", 31337 + i, 65536).tobytes() for i in range(2)]
    want = [oracle_lib.raw_compress(b) for b in blocks]
    for it in range(300):
        got = compress_many(blocks)
        assert got == want, it
    mixed = [fuzzgen.gen(k, 99 + i, n).tobytes() for i, (k, n) in enumerate(
        [("text_mtg", 65536), ("words", 300), ("lz", 5000), ("text_mtg", 65536), ("alpha2", 1000), ("runs", 40000)] * 4)]
    want = [oracle_lib.raw_compress(b) for b in mixed]
    for it in range(20):
        assert compress_many(mixed) == want, it
    rnd = [fuzzgen.gen("random", 555 + i, 65536).tobytes() for i in range(300)]
    got = compress_many(rnd)
    for i in range(0, 300, 11):
        assert got[i] == oracle_lib.raw_compress(rnd[i]), i


def test_full_size_c3_one_gib_tiled_random_ascii():
    # BASELINE.json configs[2]: every 64 KiB window of the tiling is incompressible -> one literal,
    # 3 (varint) + 3 (tag) + 65536 bytes
    _full_size_round_trip("c3", 16384, expect_len=65542)


def test_full_size_text_and_low_entropy():
    lens = _full_size_round_trip("text", 4096)
    assert 2.0 < 65536 * len(lens) / lens.sum() < 2.6
    lens = _full_size_round_trip("c5", 4096)
    assert 65536 * len(lens) / lens.sum() >= 8.0  # BASELINE.json configs[4]: >= 8x compressible
    _full_size_round_trip("mix", 3072)


def test_full_size_c4_per_gpu_share_mix_and_checksum_of_checksums():
    """BASELINE.json configs[3]: one GPU's share of the 16 GiB run = 32 768 blocks (2 GiB) of the
    c3/text/c5 mix.  Round trip of every block, sampled blocks == oracle, and a size-independent
    property: the multiset of compressed blocks depends only on the blocks, so two halves compressed
    in separate launches give the same per-block lengths and the same checksum of checksums as one
    launch over everything."""
    import torch
    lens = _full_size_round_trip("mix", 32768)
    import bench
    P = pkg()
    dev = torch.device("cuda:0")

    def digest(first, count):
        src = bench.make_input("mix", first, count, dev)
        dst, ol = P.compress_blocks(src, 65536, count)
        torch.cuda.synchronize()
        d = dst.reshape(count, P.SLOT_STRIDE)
        # per-block checksum on the GPU: sum of (byte * (position + 1)) over the used bytes, mod 2^61-1-ish in int64
        pos = torch.arange(1, P.SLOT_STRIDE + 1, device=dev, dtype=torch.int64)
        used = pos[None, :] <= ol[:, None].to(torch.int64)
        cs = torch.zeros(count, dtype=torch.int64, device=dev)
        for c0 in range(0, count, 2048):  # bounded temporaries
            sl = slice(c0, min(count, c0 + 2048))
            cs[sl] = ((d[sl].to(torch.int64) * pos[None, :]) * used[sl]).sum(dim=1)
        return ol.cpu().numpy(), cs.cpu().numpy()

    la, ca = digest(0, 8192)
    lb, cb = digest(8192, 8192)
    lw, cw = digest(0, 16384)
    assert np.array_equal(np.concatenate([la, lb]), lw)
    assert np.array_equal(np.concatenate([ca, cb]), cw)
    assert np.array_equal(lw, lens[:16384])
    assert int(np.bitwise_xor.reduce(cw)) == int(np.bitwise_xor.reduce(np.concatenate([ca, cb])))


def test_all_mtg_cards_48_blocks_bit_exact():
    # BASELINE.json configs[1]
    mtg = fuzzgen.corpus("mtg")
    blocks = [mtg[o:o + 65536] for o in range(0, len(mtg), 65536)]
    assert len(blocks) == 48
    got = compress_many(blocks)
    assert sum(len(g) for g in got) == 1334376
    for g, b in zip(got, blocks):
        assert g == oracle_lib.raw_compress(b)


def test_device_api_is_graph_capturable():
    """After one warm-up call on the stream (which creates the workspace), a compress launch -- the
    sampling kernel and both compress kernels -- records into a hipGraph and replays on new data in the
    same buffers, whatever kind of data the replay finds there: incompressible (the single kernel works),
    text (the team kernel), a mix (both, block by block)."""
    import torch
    P = pkg()
    nb = 2200  # above the auto threshold (4 x CUs): the captured launch includes the persistent kernel
    st = torch.cuda.Stream()
    src = torch.zeros(nb * 65536, dtype=torch.uint8, device="cuda")
    dst = torch.empty(nb * P.SLOT_STRIDE, dtype=torch.uint8, device="cuda")
    out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
    mtg = np.frombuffer(fuzzgen.corpus("mtg"), dtype=np.uint8)
    rnd = np.frombuffer(fuzzgen.corpus("random_ascii"), dtype=np.uint8)

    def fill(kind):
        base = mtg if kind == "text" else rnd
        reps = nb * 65536 // base.size + 2
        host = np.tile(base, reps)[7:7 + nb * 65536]
        src.copy_(torch.from_numpy(host.copy()))
        return host

    with torch.cuda.stream(st):
        fill("text")
        P.compress_blocks(src, 65536, nb, dst=dst, out_len=out_len, stream=st)  # warm-up: allocates
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            P.compress_blocks(src, 65536, nb, dst=dst, out_len=out_len, stream=torch.cuda.current_stream())
    def fill_mixed():
        # every third block text, the others incompressible: the replayed launch is ROUTED block by block (round 5:
        # the launch state, the queue's counters and the sample's arrival counter reset themselves in every replay)
        host = np.tile(rnd, nb * 65536 // rnd.size + 2)[7:7 + nb * 65536].copy().reshape(nb, 65536)
        text = np.tile(mtg, nb * 65536 // mtg.size + 2)[11:11 + nb * 65536].reshape(nb, 65536)
        host[0::3] = text[0::3]
        host = host.reshape(-1)
        src.copy_(torch.from_numpy(host))
        return host

    for kind in ("rnd", "text", "mixed", "rnd", "mixed", "mixed", "text"):
        host = fill_mixed() if kind == "mixed" else fill(kind)
        torch.cuda.synchronize()
        out_len.zero_()
        g.replay()
        torch.cuda.synchronize()
        ol = out_len.cpu().numpy()
        assert ((ol > 0) & (ol < 80000)).all(), (kind, int((ol > 0).sum()))
        checked = (0, 1, 2, 3, nb // 2, nb // 2 + 1, nb // 2 + 2, nb - 2, nb - 1) if kind == "mixed" else (0, 1, nb // 2, nb - 1)
        for b in checked:
            want = oracle_lib.raw_compress(host[b * 65536:(b + 1) * 65536])
            got = dst[b * P.SLOT_STRIDE:b * P.SLOT_STRIDE + int(ol[b])].cpu().numpy().tobytes()
            assert got == want, (kind, b)
        if kind == "mixed":
            assert (ol[1::3] == 65542).all() and (ol[0::3] < 40000).all()


def test_release_thread_resources_and_reuse():
    """sz_release_thread_resources frees the host-API workspace (and the multi-wave workspace of its
    stream); the next call simply builds them again."""
    P = pkg()
    data = fuzzgen.corpus("mtg")[:300000]
    want = oracle_lib.raw_compress(data)
    for _ in range(3):
        assert P.compress(data) == want
        P.lib().sz_release_thread_resources()
    assert P.uncompress(P.compress(data)) == data


def test_corrupted_streams_never_write_out_of_bounds():
    """Bit-flipped, truncated and spliced streams: the decompressor may report an error or (when the
    damage happens to be harmless) decode something, but it must stay inside its output slot."""
    import torch
    P = pkg()
    rs = np.random.RandomState(2024)
    good = [oracle_lib.raw_compress(fuzzgen.gen(k, 50 + i, int(rs.randint(200, 65537))).tobytes())
            for i, k in enumerate(("text_mtg", "lz", "words", "runs", "random", "alpha4", "periodic", "sparse") * 4)]
    streams = []
    for i in range(1500):
        s = bytearray(good[i % len(good)])
        mode = i % 4
        if mode == 0:
            for _ in range(int(rs.randint(1, 6))):
                s[int(rs.randint(0, len(s)))] ^= 1 << int(rs.randint(0, 8))
        elif mode == 1:
            s = s[:int(rs.randint(1, len(s)))]
        elif mode == 2:
            other = good[int(rs.randint(0, len(good)))]
            cut = int(rs.randint(1, len(s)))
            s = s[:cut] + other[int(rs.randint(0, len(other))):]
        else:
            pos = int(rs.randint(0, len(s)))
            s[pos:pos] = bytes(rs.randint(0, 256, size=int(rs.randint(1, 9))).astype(np.uint8))
        streams.append(bytes(s[:P.SLOT_STRIDE]))
    nb = len(streams)
    host = np.zeros((nb, P.SLOT_STRIDE), dtype=np.uint8)
    clen = np.zeros(nb, dtype=np.int32)
    for i, c in enumerate(streams):
        host[i, :len(c)] = np.frombuffer(c, dtype=np.uint8)
        clen[i] = len(c)
    stride = 65536 + 256  # slots are adjacent: a write past one slot's capacity lands in its neighbour
    dst = torch.full((nb * stride,), 0x5A, dtype=torch.uint8, device="cuda")
    d_src = torch.from_numpy(host).cuda().reshape(-1)
    d_len = torch.from_numpy(clen).cuda()
    out_len = torch.empty(nb, dtype=torch.int32, device="cuda")
    status = torch.empty(nb, dtype=torch.int32, device="cuda")
    import ctypes
    rc = P.lib().sz_uncompress_blocks_hip(ctypes.c_void_p(d_src.data_ptr()), P.SLOT_STRIDE, ctypes.c_void_p(d_len.data_ptr()),
                                          nb, ctypes.c_void_p(dst.data_ptr()), stride, ctypes.c_void_p(out_len.data_ptr()),
                                          ctypes.c_void_p(status.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    out = dst.cpu().numpy().reshape(nb, stride)
    st = status.cpu().numpy()
    ol = out_len.cpu().numpy()
    assert set(np.unique(st)) <= {0, P.SZ_ECORRUPT, P.SZ_ENOSPC}
    assert (st != 0).sum() > nb // 2  # most of the damage is detected
    for i in range(nb):
        if st[i] == 0:
            assert oracle_lib.uncompress(streams[i]) == out[i, :ol[i]].tobytes()


def test_host_api_from_concurrent_threads():
    """include/snappy_hip.h: the host-buffer API keeps its state per host thread (one RoCC port per
    hart).  Four threads compress and decompress different inputs at the same time."""
    import threading
    P = pkg()
    mtg = fuzzgen.corpus("mtg")
    inputs = [mtg[:300000], fuzzgen.gen("lz", 71, 250000).tobytes(), fuzzgen.corpus("random_ascii") * 2,
              fuzzgen.gen("words", 72, 200001).tobytes()]
    want = [oracle_lib.raw_compress(d) for d in inputs]
    errors = []

    def worker(i):
        try:
            for _ in range(6):
                got = P.compress(inputs[i])
                if got != want[i]:
                    errors.append("thread %d: compress differs" % i)
                if P.uncompress(got) != inputs[i]:
                    errors.append("thread %d: round trip differs" % i)
                # the RoCC length register is per thread too
                P.rocc_cmd(P.ROCC_SET_LENGTH, 1000 + i)
            P.lib().sz_release_thread_resources()
        except Exception as e:  # noqa: BLE001 - reported below
            errors.append("thread %d: %r" % (i, e))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(len(inputs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_two_streams_in_flight_at_once():
    """Each (device, stream) owns its multi-wave workspace (tables, block queue): launches on two
    streams may overlap on the GPU without sharing any of it."""
    import torch
    import bench
    P = pkg()
    dev = torch.device("cuda:0")
    nb = 3072
    src_a = bench.make_input("text", 0, nb, dev)
    src_b = bench.make_input("mix", 0, nb, dev)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for it in range(3):
        with torch.cuda.stream(sa):
            da, la = P.compress_blocks(src_a, 65536, nb, stream=sa)
        with torch.cuda.stream(sb):
            db, lb = P.compress_blocks(src_b, 65536, nb, stream=sb)
        outs.append((da, la, db, lb))
    torch.cuda.synchronize()
    rs = np.random.RandomState(5)
    picks = rs.choice(nb, size=24, replace=False)
    for da, la, db, lb in outs:
        for src, d, l in ((src_a, da, la), (src_b, db, lb)):
            ln = l.cpu().numpy()
            dn = d.reshape(nb, P.SLOT_STRIDE)
            for b in picks:
                blk = src[int(b) * 65536:(int(b) + 1) * 65536].cpu().numpy().tobytes()
                assert dn[int(b), :int(ln[b])].cpu().numpy().tobytes() == oracle_lib.raw_compress(blk), int(b)
        # every block of every launch: same lengths as the first launch on that stream
        assert torch.equal(la, outs[0][1]) and torch.equal(lb, outs[0][3])


def test_pack_reports_error_markers_instead_of_following_them():
    """A block the compress kernel refused (in_len > 64 KiB -> out_len 0xffffffff) must not turn into a
    4 GiB gather: the pack step skips it and flags the total."""
    import torch
    P = pkg()
    nb = 6
    blocks = [fuzzgen.gen("text_mtg", 900 + i, 65536).tobytes() for i in range(nb)]
    src = torch.from_numpy(np.frombuffer(b"".join(blocks), dtype=np.uint8).copy()).cuda()
    in_len = torch.tensor([65536, 65536, 70000, 65536, 0, 65536], dtype=torch.int32, device="cuda")
    slots, lens = P.compress_blocks(src, 65536, nb, in_len=in_len, varint=False)
    torch.cuda.synchronize()
    ln = lens.cpu().numpy().astype(np.uint32)
    assert ln[2] == 0xffffffff and ln[4] == 0
    packed, offs = P.pack_stream(slots, P.SLOT_STRIDE, lens, nb, base=3)
    torch.cuda.synchronize()
    o = offs.cpu().numpy().astype(np.uint64)
    assert o[nb] == np.uint64(0xffffffffffffffff)
    # the good blocks are still packed back to back (the refused one counts as empty)
    assert o[0] == 3 and o[1] == 3 + ln[0] and o[2] == o[1] + ln[1] and o[3] == o[2] and o[4] == o[3] + ln[3]
    body0 = packed[3:3 + int(ln[0])].cpu().numpy().tobytes()
    assert body0 == oracle_lib.raw_compress(blocks[0])[3:]


def test_release_stream_resources_and_reuse():
    """The per-stream multi-wave workspace can be dropped and comes back on the next call."""
    import torch
    P = pkg()
    blocks = [fuzzgen.gen("text_mtg", 77 + i, 65536).tobytes() for i in range(6)]
    want = [oracle_lib.raw_compress(b) for b in blocks]
    src = torch.from_numpy(np.frombuffer(b"".join(blocks), dtype=np.uint8).copy()).cuda()
    st = torch.cuda.Stream()
    free0 = torch.cuda.mem_get_info()[0]
    for rnd in range(3):
        os.environ["SZ_KERNEL"] = "dual"
        try:
            with torch.cuda.stream(st):
                dst, ol = P.compress_blocks(src, 65536, 6, stream=st)
            st.synchronize()
        finally:
            os.environ.pop("SZ_KERNEL", None)
        ln = ol.cpu().numpy()
        out = dst.reshape(6, P.SLOT_STRIDE).cpu().numpy()
        assert [out[i, :ln[i]].tobytes() for i in range(6)] == want
        assert P.lib().sz_release_stream_resources(ctypes.c_void_p(st.cuda_stream)) == 0
    del dst, ol
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (32 << 20)  # the 64 MiB workspace is not left behind


def _varint(n):
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7f) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def _literal(payload):
    n = len(payload) - 1
    if n < 60:
        return bytes([n << 2]) + payload
    nb = 1 if n < 256 else 2 if n < 65536 else 3 if n < (1 << 24) else 4
    return bytes([(59 + nb) << 2]) + n.to_bytes(nb, "little") + payload


@pytest.mark.parametrize("seg_shift", [None, "13", "14", "15", "16"])
def test_large_single_stream_takes_the_indexed_path_and_agrees_with_the_serial_decoder(seg_shift, monkeypatch):
    """sz_uncompress on one stream of many fragments: fragment offsets are found in parallel
    (snappy_stream_index.hip) and the fragments decoded one wave each; streams that are valid but not
    made of 64 KiB fragments, and damaged streams, must come out exactly as from the serial decoder
    (SZ_SERIAL_UNCOMPRESS=1) and the oracle."""
    if seg_shift:
        monkeypatch.setenv("SZ_TEST_SEG_SHIFT", seg_shift)  # 8 .. 64 KiB segments (default here: 4 KiB)
    P = pkg()
    mtg = fuzzgen.corpus("mtg")
    rs = np.random.RandomState(99)
    cases = {
        "text 40 fragments + ragged tail": (mtg * 2)[:65536 * 40 + 12345],
        "exact multiple of 64 KiB": fuzzgen.gen("lz", 5, 65536 * 8).tobytes(),
        "incompressible": fuzzgen.corpus("random_ascii") * 9,
        "mixed": b"".join(fuzzgen.gen(k, 300 + i, 65536).tobytes() for i, k in enumerate(("random", "text_mtg", "same", "words", "runs", "sparse", "alpha2", "periodic"))),
    }
    for name, data in cases.items():
        comp = oracle_lib.raw_compress(data)
        assert P.uncompress(comp) == data, name
    # valid streams that no 64 KiB-fragment compressor would write
    blob = rs.randint(0, 256, size=300000, dtype=np.uint8).tobytes()
    odd1 = _varint(300000) + _literal(blob[:100000]) + _literal(blob[100000:100001]) + _literal(blob[100001:])
    # copies that reach back across a 64 KiB boundary (offset 65535 at output position 70000)
    base = rs.randint(0, 256, size=70000, dtype=np.uint8).tobytes()
    tail = bytes(base[70000 - 65535 + i] for i in range(64))
    odd2_data = base + tail + base[:200000 - 70064] * 3
    odd2_data = odd2_data[:270000]
    odd2 = _varint(len(odd2_data)) + _literal(base) + bytes([(63 << 2) | 2]) + (65535).to_bytes(2, "little") + _literal(odd2_data[70064:])
    for name, comp, want in (("literals across boundaries", odd1, blob), ("copy across a boundary", odd2, odd2_data)):
        assert oracle_lib.uncompress(comp) == want, name
        assert P.uncompress(comp) == want, name
    # damage: the verdict and the bytes must not depend on the path
    good = oracle_lib.raw_compress((mtg * 2)[:65536 * 24 + 777])
    for i in range(40):
        s = bytearray(good)
        mode = i % 4
        if mode == 0:
            for _ in range(3):
                s[int(rs.randint(5, len(s)))] ^= 1 << int(rs.randint(0, 8))
        elif mode == 1:
            s = s[:int(rs.randint(len(s) // 2, len(s)))]
        elif mode == 2:
            pos = int(rs.randint(5, len(s)))
            s[pos:pos] = bytes(rs.randint(0, 256, size=3, dtype=np.uint8))
        else:
            pos = int(rs.randint(5, len(s) - 10))
            del s[pos:pos + int(rs.randint(1, 6))]
        s = bytes(s)

        def run():
            try:
                return ("ok", P.uncompress(s))
            except P.SnappyHipError as e:
                return ("err", e.code)
        got = run()
        monkeypatch.setenv("SZ_SERIAL_UNCOMPRESS", "1")
        ref = run()
        monkeypatch.delenv("SZ_SERIAL_UNCOMPRESS")
        assert got == ref, (i, mode, got[0], ref[0])
        try:
            want = ("ok", oracle_lib.uncompress(s))
        except Exception:
            want = ("err", None)
        assert got[0] == want[0] and (got[0] == "err" or got[1] == want[1]), (i, mode)


def test_device_stream_decode_without_side_information():
    """sz_uncompress_stream_hip: one raw stream in HBM -> fragment index on the GPU -> one wave per
    fragment; all three status words zero exactly when the stream is made of 64 KiB fragments and intact."""
    import torch
    P = pkg()
    mtg = fuzzgen.corpus("mtg")
    data = (mtg * 3)[:65536 * 100 + 4321]
    comp = oracle_lib.raw_compress(data)
    body = 4
    assert P.uncompressed_length(comp) == len(data)

    def run(c, ulen):
        buf = torch.zeros(len(c) + 64, dtype=torch.uint8, device="cuda")  # readable past n
        buf[:len(c)] = torch.from_numpy(np.frombuffer(c, dtype=np.uint8).copy()).cuda()
        out, st = P.uncompress_stream(buf, len(c), body, ulen)
        torch.cuda.synchronize()
        return out, st.cpu().numpy().tolist()

    out, st = run(comp, len(data))
    assert st == [0, 0, 0]
    assert out.cpu().numpy().tobytes() == data
    # flipped bits: either reported, or (a payload byte, a still-valid tag) decoded exactly as the oracle does
    rs = np.random.RandomState(17)
    reported = 0
    for _ in range(12):
        s = bytearray(comp)
        for _ in range(2):
            s[int(rs.randint(body, len(s)))] ^= 1 << int(rs.randint(0, 8))
        out, st = run(bytes(s), len(data))
        try:
            want = oracle_lib.uncompress(bytes(s))
        except ValueError:
            want = None
        if st == [0, 0, 0]:
            assert want is not None and out.cpu().numpy().tobytes() == want
        else:
            reported += 1
    assert reported >= 1
    # a valid stream that is not fragment-aligned: flagged as such (parse fine, boundary flag set)
    blob = np.random.RandomState(3).randint(0, 256, size=300000, dtype=np.uint8).tobytes()
    odd = _varint(300000) + _literal(blob[:100000]) + _literal(blob[100000:])
    buf_body = len(_varint(300000))
    bufc = torch.zeros(len(odd) + 64, dtype=torch.uint8, device="cuda")
    bufc[:len(odd)] = torch.from_numpy(np.frombuffer(odd, dtype=np.uint8).copy()).cuda()
    out, st = P.uncompress_stream(bufc, len(odd), buf_body, 300000)
    torch.cuda.synchronize()
    st = st.cpu().numpy().tolist()
    assert st[0] == 0 and st[1] != 0


@pytest.mark.timeout(300, method="thread")
def test_stream_index_beyond_two_to_the_31_bytes():
    """Round 6: the fragment index takes streams up to SZ_STREAM_INDEX_MAX_BYTES (2^32 - 256 KiB; before: 2^31 - 1, and
    2 GiB of incompressible data compress to 2 147 582 000 bytes: sz_uncompress fell to the serial decoder, 6.7 GB/s).
    A stream of 2.6e9 bytes -- 2 GiB of tiled randomASCII followed by 1 GiB of text, so that both the runs pass and the
    segment maps work behind position 2^31 --, compressed and packed on the device, decoded from its bytes alone:
    all three status words zero, output == input; and one byte beyond the limit is refused."""
    import torch
    sys_path_bench()
    import bench
    P = pkg()
    dev = torch.device("cuda:0")
    nb_a, nb_b = 32768, 16384
    src = torch.cat([bench.make_input("c3", 0, nb_a, dev), bench.make_input("text", 7, nb_b, dev)])
    n = int(src.numel())
    nb = nb_a + nb_b
    hdr = _varint(n)
    slots, lens, _ = P.compress_stream(src, n)
    packed = torch.zeros(len(hdr) + int(lens.to(torch.int64).sum().item()) + 64, dtype=torch.uint8, device=dev)
    packed[:len(hdr)] = torch.tensor(list(hdr), dtype=torch.uint8, device=dev)
    _, offs = P.pack_stream(slots, P.SLOT_STRIDE, lens, nb, base=len(hdr), packed=packed)
    total = int(offs[nb].item())
    del slots
    assert total > (1 << 31) + (1 << 28)
    back, st = P.uncompress_stream(packed, total, len(hdr), n)
    torch.cuda.synchronize()
    assert st.cpu().numpy().tolist() == [0, 0, 0]
    assert bool(torch.equal(back, src))
    # the limit itself is an argument error, before anything is launched
    rc = P.lib().sz_uncompress_stream_hip(packed.data_ptr(), 0xfffc0000 + 1, len(hdr), n, back.data_ptr(), packed.data_ptr(), 1 << 40,
                                          st.data_ptr(), None)  # (argument checks come first: nothing is launched)
    assert rc == P.SZ_EINVAL


def sys_path_bench():
    import sys
    if build_util.ROOT not in sys.path:
        sys.path.insert(0, build_util.ROOT)


def test_streams_of_big_elements_go_through_the_runs_pass():
    """Pass 0 of the stream index (one wave following runs of equal-sized big elements) covers a stream whose
    fragments are one literal each; a stream it cannot cover falls to the segment passes.  Either way the decode
    equals the input, and damage is reported or decoded exactly as the oracle decodes it."""
    import torch
    P = pkg()
    rs = np.random.RandomState(99)
    noise = rs.randint(0, 256, size=65536 * 300 + 12345, dtype=np.uint8).tobytes()
    mtg = fuzzgen.corpus("mtg")

    def run(c, ulen, body):
        buf = torch.zeros(len(c) + 64, dtype=torch.uint8, device="cuda")
        buf[:len(c)] = torch.from_numpy(np.frombuffer(c, dtype=np.uint8).copy()).cuda()
        out, st = P.uncompress_stream(buf, len(c), body, ulen)
        torch.cuda.synchronize()
        return out.cpu().numpy().tobytes(), st.cpu().numpy().tolist()

    def body_of(c):
        b = 1
        while c[b - 1] & 0x80:
            b += 1
        return b

    mixed = noise[:65536 * 150] + mtg[:65536 * 2] + noise[65536 * 150:]
    for data in (noise, noise[:65536 * 257], mixed, noise[:65536 + 1]):
        comp = oracle_lib.raw_compress(data)
        out, st = run(comp, len(data), body_of(comp))
        assert st == [0, 0, 0] and out == data
    comp = oracle_lib.raw_compress(noise)
    body = body_of(comp)
    reported = 0
    for k in range(10):
        s = bytearray(comp)
        if k < 5:  # a literal's tag or length bytes (every fragment starts 65 539 bytes after the one before)
            s[body + 65539 * int(rs.randint(0, 300)) + int(rs.randint(0, 3))] ^= 1 << int(rs.randint(0, 8))
        else:
            s[int(rs.randint(body, len(s)))] ^= 1 << int(rs.randint(0, 8))
        out, st = run(bytes(s), len(noise), body)
        try:
            want = oracle_lib.uncompress(bytes(s))
        except ValueError:
            want = None
        if st == [0, 0, 0]:
            assert want is not None and out == want
        else:
            reported += 1
    assert reported >= 3
    out, st = run(comp[:len(comp) - 70000], len(noise), body)  # truncated
    assert st != [0, 0, 0]


def test_two_host_threads_launching_on_the_same_stream():
    """The multi-wave kernel's workspace (sample counters, block queue) belongs to (device, stream): two
    host threads that launch on the SAME stream -- here the NULL stream, the default of the binding --
    each with enough blocks for that kernel, must not interleave their launch sequences (a queue
    counter reset or advanced by the other call would skip blocks and leave their lengths unwritten).
    Every block of every call is compared with the oracle."""
    import sys
    import threading
    import torch
    sys.path.insert(0, build_util.ROOT)
    import bench
    P = pkg()
    dev = torch.device("cuda:0")
    L = P.lib()
    nb = 2304  # >= 2 * (4 * 256 CUs) blocks: the multi-wave kernel in auto mode
    null_stream = ctypes.c_void_p(0)
    jobs = []
    for i, kind in enumerate(("text", "mix")):
        src = bench.make_input(kind, 1000 * i, nb, dev)
        dst = torch.zeros(nb * P.SLOT_STRIDE, dtype=torch.uint8, device=dev)
        ol = torch.full((nb,), -7, dtype=torch.int32, device=dev)
        jobs.append((kind, src, dst, ol))
    torch.cuda.synchronize()
    errors = []

    def run(job):
        kind, src, dst, ol = job
        for _ in range(6):
            rc = L.sz_compress_blocks_hip(ctypes.c_void_p(src.data_ptr()), 65536, None, nb, ctypes.c_void_p(dst.data_ptr()),
                                          P.SLOT_STRIDE, ctypes.c_void_p(ol.data_ptr()), 1, null_stream)
            if rc != 0:
                errors.append((kind, rc))

    ts = [threading.Thread(target=run, args=(j,)) for j in jobs]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for kind, src, dst, ol in jobs:
        assert int((ol < 0).sum().item()) == 0, kind
        parity_util.assert_all_blocks_match_oracle(src, dst, ol, nb, P.SLOT_STRIDE, what="two threads, " + kind)
