"""ctypes binding of oracle/liboracle_snappy.so (CPU restatement of Snappy 1.1.8).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
BLOCK = 65536

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle_snappy.so"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(ORACLE_DIR, "liboracle_snappy.so")
    src = os.path.join(ORACLE_DIR, "snappy_oracle.c")
    if not os.path.exists(path) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path)):
        build()
    L = ctypes.CDLL(path)
    u8p = ctypes.c_void_p
    L.oracle_max_compressed_length.argtypes = [ctypes.c_size_t]
    L.oracle_max_compressed_length.restype = ctypes.c_size_t
    L.oracle_compress_fragment.argtypes = [u8p, ctypes.c_size_t, u8p]
    L.oracle_compress_fragment.restype = ctypes.c_size_t
    L.oracle_raw_compress.argtypes = [u8p, ctypes.c_size_t, u8p]
    L.oracle_raw_compress.restype = ctypes.c_size_t
    L.oracle_uncompress.argtypes = [u8p, ctypes.c_size_t, u8p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.oracle_uncompress.restype = ctypes.c_int
    L.oracle_get_uncompressed_length.argtypes = [u8p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32)]
    L.oracle_get_uncompressed_length.restype = ctypes.c_size_t
    L.oracle_compress_blocks.argtypes = [u8p, ctypes.c_size_t, u8p, ctypes.c_uint32, u8p, ctypes.c_size_t,
                                         u8p, ctypes.c_int, ctypes.c_int]
    L.oracle_compress_blocks.restype = ctypes.c_int
    L.oracle_time_compress_blocks.argtypes = [u8p, ctypes.c_size_t, ctypes.c_uint32, u8p, ctypes.c_size_t, u8p,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    L.oracle_time_compress_blocks.restype = ctypes.c_int
    L.oracle_time_external_blocks.argtypes = [ctypes.c_void_p, u8p, ctypes.c_size_t, ctypes.c_uint32, u8p, ctypes.c_size_t,
                                              u8p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    L.oracle_time_external_blocks.restype = ctypes.c_int
    L.oracle_fnv1a64.argtypes = [u8p, ctypes.c_size_t]
    L.oracle_fnv1a64.restype = ctypes.c_uint64
    _lib = L
    return L


def _as_u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def max_compressed_length(n):
    return lib().oracle_max_compressed_length(n)


def raw_compress(data):
    """varint32(n) + fragment bodies == snappy::Compress output (bytes)."""
    a = _as_u8(data)
    out = np.empty(max_compressed_length(a.size), dtype=np.uint8)
    k = lib().oracle_raw_compress(a.ctypes.data, a.size, out.ctypes.data)
    return out[:k].tobytes()


def compress_fragment(data):
    a = _as_u8(data)
    assert a.size <= BLOCK
    out = np.empty(max_compressed_length(a.size), dtype=np.uint8)
    k = lib().oracle_compress_fragment(a.ctypes.data, a.size, out.ctypes.data)
    return out[:k].tobytes()


def uncompress(comp):
    a = _as_u8(comp)
    ulen = ctypes.c_uint32(0)
    if lib().oracle_get_uncompressed_length(a.ctypes.data, a.size, ctypes.byref(ulen)) == 0:
        raise ValueError("bad varint preamble")
    out = np.empty(max(ulen.value, 1), dtype=np.uint8)
    n = ctypes.c_size_t(0)
    rc = lib().oracle_uncompress(a.ctypes.data, a.size, out.ctypes.data, ulen.value, ctypes.byref(n))
    if rc != 0:
        raise ValueError("malformed snappy stream (code %d)" % rc)
    return out[:n.value].tobytes()


def compress_blocks(src, in_stride, n_blocks, in_len=None, dst_stride=None, per_block_varint=True, threads=1):
    """Batch form; returns (dst uint8[n_blocks, dst_stride], out_len uint32[n_blocks])."""
    a = _as_u8(src)
    if dst_stride is None:
        dst_stride = max_compressed_length(BLOCK) + 6
    dst = np.empty((n_blocks, dst_stride), dtype=np.uint8)
    out_len = np.zeros(n_blocks, dtype=np.uint32)
    il = None
    if in_len is not None:
        il = np.ascontiguousarray(in_len, dtype=np.uint32)
    rc = lib().oracle_compress_blocks(a.ctypes.data, in_stride, il.ctypes.data if il is not None else None,
                                      n_blocks, dst.ctypes.data, dst_stride, out_len.ctypes.data,
                                      1 if per_block_varint else 0, threads)
    if rc != 0:
        raise RuntimeError("oracle_compress_blocks rc=%d" % rc)
    return dst, out_len


def time_compress_blocks(src, in_stride, n_blocks, threads, reps, per_block_varint=True):
    """Seconds for `reps` passes of the oracle over n_blocks fixed-size blocks on `threads` pinned
    pthreads (buffers allocated here, outside the timed region; one warm-up pass).  CPU baseline."""
    a = _as_u8(src)
    dst_stride = max_compressed_length(in_stride) + 6
    dst = np.empty((n_blocks, dst_stride), dtype=np.uint8)
    out_len = np.zeros(n_blocks, dtype=np.uint32)
    sec = ctypes.c_double(0.0)
    rc = lib().oracle_time_compress_blocks(a.ctypes.data, in_stride, n_blocks, dst.ctypes.data, dst_stride,
                                           out_len.ctypes.data, 1 if per_block_varint else 0, threads, reps,
                                           ctypes.byref(sec))
    if rc != 0:
        raise RuntimeError("oracle_time_compress_blocks rc=%d" % rc)
    return sec.value, out_len


def time_external_blocks(fn_addr, src, in_stride, n_blocks, threads, reps):
    """Seconds for `reps` passes of an external compressor with libsnappy's C signature (fn_addr = the
    address of snappy_compress) in the same pinned-thread harness as time_compress_blocks."""
    a = _as_u8(src)
    dst_stride = max_compressed_length(in_stride) + 6
    dst = np.empty((n_blocks, dst_stride), dtype=np.uint8)
    out_len = np.zeros(n_blocks, dtype=np.uint32)
    sec = ctypes.c_double(0.0)
    rc = lib().oracle_time_external_blocks(fn_addr, a.ctypes.data, in_stride, n_blocks, dst.ctypes.data, dst_stride,
                                           out_len.ctypes.data, threads, reps, ctypes.byref(sec))
    if rc != 0:
        raise RuntimeError("oracle_time_external_blocks rc=%d" % rc)
    return sec.value, out_len


def fnv1a64(data):
    a = _as_u8(data)
    return lib().oracle_fnv1a64(a.ctypes.data, a.size)
