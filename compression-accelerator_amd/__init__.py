"""compression-accelerator_amd -- MI355X-native Snappy block compressor (host-side Python plumbing).

The product is the C-ABI shared library csrc/libsnappy_hip.so (include/snappy_hip.h); this module
is a thin ctypes binding used by tests/, bench.py and __graft_entry__.py.  PyTorch is used only
for device memory, streams and torch.distributed -- no torch types cross the C ABI.

There is no CPU fallback: if the library is not built this module raises at import, and without
a GPU every compute call raises SnappyHipError(SZ_ENODEV).

(The directory name has a hyphen; load it with `load_package()` from tests/pkg_loader.py or
importlib -- see __graft_entry__.py.)
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libsnappy_hip.so")

BLOCK_SIZE = 65536
SLOT_STRIDE = 76544
FLAG_BLOCK_VARINT = 1
FLAG_HW_COMPAT = 2  # the reference accelerator's own element stream instead of software Snappy's (include/snappy_hip.h)

SZ_OK, SZ_EINVAL, SZ_ENOSPC, SZ_EHIP, SZ_ENODEV, SZ_ECORRUPT = 0, -22, -28, -5, -19, -74

ROCC_CHANNEL, ROCC_COMPRESS, ROCC_UNCOMPRESS, ROCC_SET_LENGTH = 3, 0, 1, 2

# every symbol include/snappy_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = (
    "sz_max_compressed_length", "sz_compress_blocks_hip", "sz_compress_stream_hip", "sz_pack_stream_hip",
    "sz_uncompress_blocks_hip", "sz_uncompress_packed_hip", "sz_uncompress_stream_work_bytes",
    "sz_uncompress_stream_hip", "sz_compress", "sz_compress_begin", "sz_compress_finish", "sz_uncompress",
    "sz_get_uncompressed_length",
    "sz_rocc_cmd", "rocc_cmd", "sz_rocc_last_error", "sz_release_thread_resources",
    "sz_release_stream_resources", "sz_shard_range", "sz_stream_offsets", "sz_device_count",
    "sz_last_hip_error", "sz_strerror", "sz_version", "sz_profile_next_launch",
)


class SnappyHipError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        msg = "%s failed: %s (code %d, hip error %d)" % (
            what, _lib.sz_strerror(code).decode() if _lib else "?", code,
            _lib.sz_last_hip_error() if _lib else -1)
        super().__init__(msg)


def build(verbose=False):
    """Compiles csrc/ for gfx950 with hipcc (in-tree .so)."""
    cmd = ["make", "-C", CSRC] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libsnappy_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C compression-accelerator_amd/csrc` (there is no CPU fallback)")
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, u32, u64, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
    L.sz_max_compressed_length.argtypes = [sz]
    L.sz_max_compressed_length.restype = sz
    L.sz_compress_blocks_hip.argtypes = [vp, sz, vp, u32, vp, sz, vp, u32, vp]
    L.sz_compress_blocks_hip.restype = i32
    L.sz_compress_stream_hip.argtypes = [vp, sz, vp, sz, vp, u32, vp]
    L.sz_compress_stream_hip.restype = i32
    L.sz_pack_stream_hip.argtypes = [vp, sz, vp, u32, u64, vp, vp, vp]
    L.sz_pack_stream_hip.restype = i32
    L.sz_uncompress_blocks_hip.argtypes = [vp, sz, vp, u32, vp, sz, vp, vp, vp]
    L.sz_uncompress_blocks_hip.restype = i32
    L.sz_uncompress_packed_hip.argtypes = [vp, vp, u32, u64, vp, vp, vp, vp]
    L.sz_uncompress_packed_hip.restype = i32
    L.sz_uncompress_stream_work_bytes.argtypes = [sz, u64]
    L.sz_uncompress_stream_work_bytes.restype = sz
    L.sz_uncompress_stream_hip.argtypes = [vp, sz, sz, u64, vp, vp, sz, vp, vp]
    L.sz_uncompress_stream_hip.restype = i32
    L.sz_compress.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz)]
    L.sz_compress.restype = i32
    L.sz_compress_begin.argtypes = [vp, sz]
    L.sz_compress_begin.restype = i32
    L.sz_compress_finish.argtypes = [vp, sz, ctypes.POINTER(sz)]
    L.sz_compress_finish.restype = i32
    L.sz_uncompress.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz)]
    L.sz_uncompress.restype = i32
    L.sz_get_uncompressed_length.argtypes = [vp, sz, ctypes.POINTER(sz)]
    L.sz_get_uncompressed_length.restype = i32
    L.sz_rocc_cmd.argtypes = [i32, i32, u64, u64, ctypes.POINTER(u64)]
    L.sz_rocc_cmd.restype = i32
    L.rocc_cmd.argtypes = [i32, i32, u64, u64]
    L.rocc_cmd.restype = u64
    L.sz_rocc_last_error.restype = i32
    L.sz_release_thread_resources.restype = None
    L.sz_release_stream_resources.argtypes = [vp]
    L.sz_release_stream_resources.restype = i32
    L.sz_shard_range.argtypes = [u64, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    L.sz_shard_range.restype = i32
    L.sz_stream_offsets.argtypes = [vp, u64, u64, u64, vp, ctypes.POINTER(u64)]
    L.sz_stream_offsets.restype = i32
    L.sz_device_count.restype = i32
    L.sz_last_hip_error.restype = i32
    L.sz_strerror.argtypes = [i32]
    L.sz_strerror.restype = ctypes.c_char_p
    L.sz_version.restype = ctypes.c_char_p
    L.sz_profile_next_launch.argtypes = [vp, vp, vp, vp]
    L.sz_profile_next_launch.restype = i32
    _lib = L
    return L


def _check(rc, what):
    if rc != SZ_OK:
        raise SnappyHipError(rc, what)


def max_compressed_length(n):
    return lib().sz_max_compressed_length(n)


def device_count():
    return lib().sz_device_count()


# ---------------------------------------------------------------- device-resident API (torch tensors)

def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream_handle(stream):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


def compress_blocks(src, in_stride, n_blocks, in_len=None, dst=None, dst_stride=SLOT_STRIDE, out_len=None,
                    varint=True, stream=None, hw_compat=False):
    """src: uint8 CUDA tensor holding n_blocks blocks of in_stride bytes (in_len: optional uint32 CUDA
    tensor of per-block lengths).  Returns (dst uint8[n_blocks*dst_stride], out_len uint32[n_blocks]);
    asynchronous on `stream` (default: torch's current stream)."""
    import torch
    assert src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()
    if dst is None:
        dst = torch.empty(n_blocks * dst_stride, dtype=torch.uint8, device=src.device)
    if out_len is None:
        out_len = torch.empty(n_blocks, dtype=torch.int32, device=src.device)
    _check(lib().sz_compress_blocks_hip(_ptr(src), in_stride, _ptr(in_len), n_blocks, _ptr(dst), dst_stride,
                                        _ptr(out_len), (FLAG_BLOCK_VARINT if varint else 0) | (FLAG_HW_COMPAT if hw_compat else 0),
                                        _stream_handle(stream)), "sz_compress_blocks_hip")
    return dst, out_len


def profile_next_launch(single_start=None, single_stop=None, multi_start=None, multi_stop=None):
    """Raw hipEvent_t handles (integers) to be recorded around the dominant kernels of this thread's next compress
    launch (include/snappy_hip.h: sz_profile_next_launch)."""
    _check(lib().sz_profile_next_launch(*(ctypes.c_void_p(e) if e else None
                                          for e in (single_start, single_stop, multi_start, multi_stop))),
           "sz_profile_next_launch")


def compress_stream(src, n, slots=None, slot_stride=SLOT_STRIDE, lens=None, varint=False, stream=None):
    """One contiguous n-byte CUDA buffer cut into 64 KiB fragments (ragged tail)."""
    import torch
    nb = (n + BLOCK_SIZE - 1) // BLOCK_SIZE
    if slots is None:
        slots = torch.empty(max(nb, 1) * slot_stride, dtype=torch.uint8, device=src.device)
    if lens is None:
        lens = torch.empty(max(nb, 1), dtype=torch.int32, device=src.device)
    _check(lib().sz_compress_stream_hip(_ptr(src), n, _ptr(slots), slot_stride, _ptr(lens),
                                        FLAG_BLOCK_VARINT if varint else 0, _stream_handle(stream)),
           "sz_compress_stream_hip")
    return slots, lens, nb


def pack_stream(slots, slot_stride, lens, n_blocks, base=0, packed=None, offsets=None, stream=None):
    """Device-side compaction: returns (packed uint8, offsets int64[n_blocks+1])."""
    import torch
    if offsets is None:
        offsets = torch.empty(n_blocks + 1, dtype=torch.int64, device=slots.device)
    if packed is None:
        packed = torch.empty(base + n_blocks * slot_stride, dtype=torch.uint8, device=slots.device)
    _check(lib().sz_pack_stream_hip(_ptr(slots), slot_stride, _ptr(lens), n_blocks, base, _ptr(packed),
                                    _ptr(offsets), _stream_handle(stream)), "sz_pack_stream_hip")
    return packed, offsets


def uncompress_blocks(src, src_stride, src_len, n_streams, dst_stride=BLOCK_SIZE, dst=None, stream=None):
    """Batch decode; returns (dst, out_len int32[n], status int32[n])."""
    import torch
    if dst is None:
        dst = torch.empty(n_streams * dst_stride, dtype=torch.uint8, device=src.device)
    out_len = torch.empty(n_streams, dtype=torch.int32, device=src.device)
    status = torch.empty(n_streams, dtype=torch.int32, device=src.device)
    _check(lib().sz_uncompress_blocks_hip(_ptr(src), src_stride, _ptr(src_len), n_streams, _ptr(dst), dst_stride,
                                          _ptr(out_len), _ptr(status), _stream_handle(stream)),
           "sz_uncompress_blocks_hip")
    return dst, out_len, status


def uncompress_packed(packed, offsets, n_blocks, total_ulen, dst=None, stream=None):
    """Parallel decode of the fragments of one packed stream (pack_stream's layout and offsets);
    returns (dst uint8[n_blocks*65536], out_len int32[n], status int32[n])."""
    import torch
    if dst is None:
        dst = torch.empty(n_blocks * BLOCK_SIZE, dtype=torch.uint8, device=packed.device)
    out_len = torch.empty(n_blocks, dtype=torch.int32, device=packed.device)
    status = torch.empty(n_blocks, dtype=torch.int32, device=packed.device)
    _check(lib().sz_uncompress_packed_hip(_ptr(packed), _ptr(offsets), n_blocks, total_ulen, _ptr(dst),
                                          _ptr(out_len), _ptr(status), _stream_handle(stream)),
           "sz_uncompress_packed_hip")
    return dst, out_len, status


def uncompress_stream(src, n, body, ulen, dst=None, stream=None):
    """One raw stream in HBM, fragments found and decoded in parallel; returns (dst, status3 int32[3]):
    all zero = decoded, else fall back to uncompress_blocks with one stream.  `src` must be readable
    8 bytes past n (allocate a little more)."""
    import torch
    if dst is None:
        dst = torch.empty(ulen, dtype=torch.uint8, device=src.device)
    wb = lib().sz_uncompress_stream_work_bytes(n, ulen)
    work = torch.empty(wb, dtype=torch.uint8, device=src.device)
    status3 = torch.empty(3, dtype=torch.int32, device=src.device)
    _check(lib().sz_uncompress_stream_hip(_ptr(src), n, body, ulen, _ptr(dst), _ptr(work), wb, _ptr(status3),
                                          _stream_handle(stream)), "sz_uncompress_stream_hip")
    return dst, status3


# ---------------------------------------------------------------------------- host-buffer API

def compress(data):
    """snappy::Compress(data) -> bytes, computed on the GPU."""
    data = bytes(data)
    cap = max_compressed_length(len(data))
    buf = ctypes.create_string_buffer(cap)
    n = ctypes.c_size_t(0)
    src = ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p)
    _check(lib().sz_compress(src, len(data), ctypes.cast(buf, ctypes.c_void_p), cap, ctypes.byref(n)), "sz_compress")
    return buf.raw[:n.value]


def uncompressed_length(comp):
    comp = bytes(comp)
    n = ctypes.c_size_t(0)
    _check(lib().sz_get_uncompressed_length(ctypes.cast(ctypes.c_char_p(comp), ctypes.c_void_p), len(comp),
                                            ctypes.byref(n)), "sz_get_uncompressed_length")
    return n.value


def uncompress(comp):
    """snappy::Uncompress(comp) -> bytes, computed on the GPU."""
    comp = bytes(comp)
    ulen = uncompressed_length(comp)
    buf = ctypes.create_string_buffer(max(ulen, 1))
    n = ctypes.c_size_t(0)
    _check(lib().sz_uncompress(ctypes.cast(ctypes.c_char_p(comp), ctypes.c_void_p), len(comp),
                               ctypes.cast(buf, ctypes.c_void_p), ulen, ctypes.byref(n)), "sz_uncompress")
    return buf.raw[:n.value]


def rocc_cmd(funct, rs1=0, rs2=0, channel=ROCC_CHANNEL):
    """One RoCC command (tests/compression.c:20-39); returns rd."""
    rd = ctypes.c_uint64(0)
    _check(lib().sz_rocc_cmd(channel, funct, rs1, rs2, ctypes.byref(rd)), "sz_rocc_cmd")
    return rd.value
