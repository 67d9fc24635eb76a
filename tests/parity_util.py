"""Full-coverage parity: EVERY block of a device-resident batch against the oracle (bytes and lengths).

TEST INFRASTRUCTURE (tests/ and bench.py's check after the timed region).  The batch is brought to the
host in chunks, the oracle (oracle/snappy_oracle.c, pthreads) compresses the same chunk, and the
comparison is vectorised: lengths equal, and every byte below the length equal."""
import os

import numpy as np

import oracle_lib

BLOCK = 65536


def host_threads():
    return max(1, len(os.sched_getaffinity(0)))


def count_mismatching_blocks(src, dst, out_len, n_blocks, slot_stride, chunk=2048, in_len=None, varint=True, block=BLOCK):
    """src: uint8 CUDA tensor [n_blocks * 65536]; dst: uint8 CUDA tensor [n_blocks * slot_stride];
    out_len: int32 CUDA tensor [n_blocks].  Returns (blocks that differ from the oracle, first bad index or -1)."""
    import torch
    threads = host_threads()
    lens_all = out_len.cpu().numpy().astype(np.int64)
    bad, first = 0, -1
    ostride = min(oracle_lib.max_compressed_length(block) + 6, slot_stride)
    chunk = max(1, chunk * BLOCK // block)
    cols = np.arange(ostride, dtype=np.int32)[None, :]
    for b0 in range(0, n_blocks, chunk):
        nb = min(chunk, n_blocks - b0)
        hin = src[b0 * block:(b0 + nb) * block].cpu().numpy()
        hout = dst[b0 * slot_stride:(b0 + nb) * slot_stride].cpu().numpy().reshape(nb, slot_stride)[:, :ostride]
        il = None if in_len is None else in_len[b0:b0 + nb].cpu().numpy().astype(np.uint32)
        want, wl = oracle_lib.compress_blocks(hin, block, nb, in_len=il, dst_stride=ostride, per_block_varint=varint,
                                              threads=threads)
        wl = wl.astype(np.int64)
        neq = wl != lens_all[b0:b0 + nb]
        mask = cols < wl[:, None]
        diff = ((hout != want) & mask).any(axis=1) | neq
        k = int(diff.sum())
        if k and first < 0:
            first = b0 + int(np.nonzero(diff)[0][0])
        bad += k
    return bad, first


def assert_all_blocks_match_oracle(src, dst, out_len, n_blocks, slot_stride, what="", **kw):
    bad, first = count_mismatching_blocks(src, dst, out_len, n_blocks, slot_stride, **kw)
    assert bad == 0, "%s: %d of %d blocks differ from the oracle (first: block %d)" % (what, bad, n_blocks, first)
