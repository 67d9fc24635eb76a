"""Buffer layouts the callers of the C ABI are free to choose (include/snappy_hip.h: plain pointers, any stride that
holds a block): source and destination bases that are not aligned to anything, input strides that are not 64 KiB,
output strides that are odd and as tight as the header allows.  The kernels align their own 16-byte traffic to the
DESTINATION (the register stream of an incompressible block, the long-match and bulk-literal copies), so every
residue of both bases is a different code path.  Every block against the oracle, through every compress kernel and
back through the GPU decompressor."""
import numpy as np
import pytest

import fuzzgen
import oracle_lib
from gpu_util import pkg

pytestmark = pytest.mark.gpu

BLOCK = 65536


@pytest.fixture(params=["auto", "single", "team", "dual"], autouse=True)
def kernel_mode(request, monkeypatch):
    if request.param == "auto":
        monkeypatch.delenv("SZ_KERNEL", raising=False)
    else:
        monkeypatch.setenv("SZ_KERNEL", request.param)
    return request.param


def _blocks(n_blocks, seed):
    """A mix of kinds and lengths: full blocks of text / incompressible / low-entropy data and ragged ones."""
    rs = np.random.RandomState(seed)
    text = fuzzgen.corpus("mtg")
    rnd = fuzzgen.corpus("random_ascii")
    out = []
    for i in range(n_blocks):
        k = i % 5
        n = BLOCK if i % 3 else int(rs.randint(1, BLOCK + 1))
        if k == 0:
            o = int(rs.randint(0, len(text) - BLOCK))
            b = text[o:o + n]
        elif k == 1:
            b = fuzzgen.tiled_block(rnd, i)[:n]
        elif k == 2:
            b = (text[:4096] * 16)[:n]
        elif k == 3:
            b = fuzzgen.gen(fuzzgen.KINDS[i % len(fuzzgen.KINDS)], seed + i, n).tobytes()
        else:
            b = bytes(rs.randint(0, 256, n, dtype=np.uint8))
        out.append(bytes(b))
    return out


def _run(blocks, src_off, dst_off, in_stride, dst_stride, varint=True):
    import torch
    P = pkg()
    nb = len(blocks)
    host = np.zeros(src_off + nb * in_stride + 64, dtype=np.uint8)
    lens = np.zeros(nb, dtype=np.int32)
    for i, b in enumerate(blocks):
        host[src_off + i * in_stride: src_off + i * in_stride + len(b)] = np.frombuffer(b, dtype=np.uint8)
        lens[i] = len(b)
    d_all = torch.from_numpy(host).cuda()
    src = d_all[src_off:]  # (a view: data_ptr() is the unaligned base)
    in_len = torch.from_numpy(lens).cuda()
    out_all = torch.full((dst_off + nb * dst_stride + 64,), 0xA5, dtype=torch.uint8, device="cuda")
    dst = out_all[dst_off:]
    assert src.data_ptr() % 16 == src_off % 16 and dst.data_ptr() % 16 == dst_off % 16
    _, out_len = P.compress_blocks(src, in_stride, nb, in_len=in_len, dst=dst, dst_stride=dst_stride, varint=varint)
    torch.cuda.synchronize()
    ol = out_len.cpu().numpy()
    got = out_all.cpu().numpy()
    # nothing in front of the first slot, nothing behind the last one
    assert (got[:dst_off] == 0xA5).all(), "bytes in front of the destination were written"
    assert (got[dst_off + nb * dst_stride:] == 0xA5).all(), "bytes behind the last slot were written"
    for i, b in enumerate(blocks):
        want = oracle_lib.raw_compress(b)
        if not varint:
            k = 1
            while want[k - 1] & 0x80:
                k += 1
            want = want[k:]
        s = dst_off + i * dst_stride
        assert ol[i] == len(want), "block %d (%d bytes): length %d, oracle %d" % (i, len(b), ol[i], len(want))
        assert got[s:s + len(want)].tobytes() == want, "block %d (%d bytes) differs from the oracle" % (i, len(b))
        # a slot's neighbours are never touched: what lies behind its length inside the slot stays as it was
        assert (got[s + len(want):s + dst_stride] == 0xA5).all(), "block %d wrote past its reported length" % i
    if varint:  # and back through the GPU decompressor, from the same unaligned slots
        back, ul, st = P.uncompress_blocks(dst, dst_stride, out_len, nb)
        torch.cuda.synchronize()
        ul = ul.cpu().numpy(); st = st.cpu().numpy(); back = back.cpu().numpy().reshape(nb, BLOCK)
        for i, b in enumerate(blocks):
            assert st[i] == 0 and ul[i] == len(b) and back[i, :len(b)].tobytes() == b, "block %d does not decode back" % i


@pytest.mark.parametrize("src_off,dst_off", [(1, 0), (0, 1), (3, 5), (7, 9), (13, 15), (15, 2), (8, 8), (4, 12)])
def test_unaligned_bases(src_off, dst_off):
    P = pkg()
    _run(_blocks(40, 100 + src_off * 16 + dst_off), src_off, dst_off, BLOCK, P.SLOT_STRIDE)


@pytest.mark.parametrize("in_stride,dst_stride", [(BLOCK + 1, 76495), (BLOCK + 13, 76501), (70001, 76544 + 7), (BLOCK, 76495)])
def test_odd_strides(in_stride, dst_stride):
    # 76 495 = sz_max_compressed_length(65536) + 5: the tightest stride the entry point accepts
    _run(_blocks(40, 7 + in_stride), 0, 0, in_stride, dst_stride)


def test_short_input_stride_and_bodies_only():
    """Blocks of at most 20 000 bytes 20 011 bytes apart, bodies without the varint, odd bases on both sides."""
    blocks = [b[:20000] for b in _blocks(48, 99)]
    _run(blocks, 5, 11, 20011, oracle_lib.max_compressed_length(20011) + 5 + 3, varint=False)


def test_large_launch_with_unaligned_bases():
    """Enough blocks for the sampled choice and the persistent kernels (2 x 4 x CUs and more), bases off by 9 and 7."""
    blocks = _blocks(64, 5)
    _run(blocks * 36, 9, 7, BLOCK, 76501)


def test_profile_events_bracket_the_dominant_kernel_once(kernel_mode):
    """sz_profile_next_launch (include/snappy_hip.h): the events handed over are recorded around the kernel that does
    the work of the NEXT launch of this thread -- and of that launch only."""
    import sys
    import torch
    from build_util import ROOT
    sys.path.insert(0, ROOT)
    import bench
    P = pkg()
    nb = 2304  # (enough blocks for the sampled choice and the persistent kernels)
    src = bench.make_input("text", 0, nb, torch.device("cuda:0"))
    hev = bench.HipEvents(2)
    try:
        multi = kernel_mode != "single"  # text: the sample sends it to the persistent kernel unless one is forced
        four = hev.ev[0]
        P.profile_next_launch(four[0], four[1], four[2], four[3])
        dst, ol = P.compress_blocks(src, BLOCK, nb)
        P.compress_blocks(src, BLOCK, nb, dst=dst, out_len=ol)  # (one shot: this launch records nothing)
        torch.cuda.synchronize()
        t = hev.ms(0, multi)
        assert t is not None and 0.05 < t < 50.0, t  # the kernel that compressed 144 MiB of text
        if kernel_mode == "auto":  # both were enqueued; the one not chosen exits at once
            other = hev.ms(0, not multi)
            assert other is not None and other < t / 4, (other, t)
        assert hev.ms(1, multi) is None  # never handed to the library: never recorded
    finally:
        hev.close()
