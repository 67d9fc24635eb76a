"""Helpers for the -m gpu tests: batch inputs through the C ABI (via the ctypes binding)."""
import numpy as np

from pkg_loader import load_package

BLOCK = 65536


def pkg():
    return load_package()


def compress_many(blocks, varint=True, device="cuda:0", hw_compat=False):
    """blocks: list of bytes-like (each <= 64 KiB).  One launch; returns list of bytes."""
    import torch
    P = pkg()
    nb = len(blocks)
    host = np.zeros((nb, BLOCK), dtype=np.uint8)
    lens = np.zeros(nb, dtype=np.int32)
    for i, b in enumerate(blocks):
        a = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b
        host[i, :a.size] = a
        lens[i] = a.size
    src = torch.from_numpy(host).to(device).reshape(-1)
    in_len = torch.from_numpy(lens).to(device)
    # poison the output so stale bytes cannot pass for correct ones
    dst = torch.full((nb * P.SLOT_STRIDE,), 0xA5, dtype=torch.uint8, device=device)
    dst, out_len = P.compress_blocks(src, BLOCK, nb, in_len=in_len, dst=dst, varint=varint, hw_compat=hw_compat)
    torch.cuda.synchronize()
    out = dst.cpu().numpy().reshape(nb, P.SLOT_STRIDE)
    ol = out_len.cpu().numpy()
    res = []
    for i in range(nb):
        assert 0 <= ol[i] <= P.SLOT_STRIDE, "block %d: bad length %d" % (i, ol[i])
        # nothing may be written past the reported length
        assert (out[i, ol[i]:] == 0xA5).all(), "block %d wrote past its reported length" % i
        res.append(out[i, :ol[i]].tobytes())
    return res
