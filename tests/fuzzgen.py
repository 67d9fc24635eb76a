"""Deterministic synthetic inputs shared by the golden generator, the oracle tests,
the GPU parity tests and bench.py.

Everything here is a pure function of (kind, seed, n); numpy's legacy RandomState is
used because its streams are frozen across numpy versions.  The corpora come from
tests/golden/data (data fixtures copied from the reference's data/ directory, see
tests/golden/README.md).
"""
import lzma
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DATA_DIR = os.path.join(GOLDEN_DIR, "data")

_cache = {}


def corpus(name):
    """fox | mtg | random_ascii  ->  bytes"""
    if name in _cache:
        return _cache[name]
    if name == "fox":
        with open(os.path.join(DATA_DIR, "fox-in-socks.txt"), "rb") as f:
            b = f.read()
    elif name == "random_ascii":
        with open(os.path.join(DATA_DIR, "randomASCII.txt"), "rb") as f:
            b = f.read()
    elif name == "mtg":
        with lzma.open(os.path.join(DATA_DIR, "all-mtg-cards.txt.xz"), "rb") as f:
            b = f.read()
    else:
        raise KeyError(name)
    _cache[name] = b
    return b


def java_random_ascii(n=100000, seed=4444):
    """Regenerates data/randomASCII.txt: scala.util.Random(4444).nextInt(93)+32 per byte
    (reference src/test/scala/dataset/DataGenerator.scala:13-16).  java.util.Random is the
    48-bit LCG seed=(seed^0x5DEECE66D); next(31)=seed>>17 after seed=seed*0x5DEECE66D+0xB;
    nextInt(bound) for non-power-of-two bound uses the rejection loop
    bits-(bits%bound)+(bound-1) < 0 in 32-bit signed arithmetic."""
    mask = (1 << 48) - 1
    s = (seed ^ 0x5DEECE66D) & mask
    out = bytearray(n)
    bound = 93
    for i in range(n):
        while True:
            s = (s * 0x5DEECE66D + 0xB) & mask
            bits = s >> 17  # next(31)
            val = bits % bound
            if bits - val + (bound - 1) < (1 << 31):
                break
        out[i] = val + 32
    return bytes(out)


KINDS = (
    "random", "alpha2", "alpha4", "periodic", "lz", "same", "words", "runs",
    "text_fox", "text_mtg", "ascii93", "sparse",
)


def gen(kind, seed, n):
    """Returns n bytes (numpy uint8 array) for a (kind, seed)."""
    rs = np.random.RandomState(seed & 0x7FFFFFFF)
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    if kind == "random":
        return rs.randint(0, 256, size=n).astype(np.uint8)
    if kind == "ascii93":
        return (rs.randint(0, 93, size=n) + 32).astype(np.uint8)
    if kind == "alpha2":
        return (rs.randint(0, 2, size=n) + 97).astype(np.uint8)
    if kind == "alpha4":
        return (rs.randint(0, 4, size=n) + 97).astype(np.uint8)
    if kind == "same":
        return np.full(n, 97 + (seed % 26), dtype=np.uint8)
    if kind == "periodic":
        period = int(rs.randint(1, 600))
        pat = rs.randint(0, 256, size=period).astype(np.uint8)
        return np.resize(pat, n)
    if kind == "sparse":
        a = np.zeros(n, dtype=np.uint8)
        k = max(1, n // 37)
        a[rs.randint(0, n, size=k)] = rs.randint(1, 256, size=k)
        return a
    if kind == "runs":
        out = np.empty(n, dtype=np.uint8)
        pos = 0
        while pos < n:
            ln = int(rs.randint(1, 200))
            out[pos:pos + ln] = rs.randint(0, 256)
            pos += ln
        return out
    if kind == "words":
        nwords = int(rs.randint(4, 400))
        words = [rs.randint(97, 123, size=int(rs.randint(1, 12))).astype(np.uint8).tobytes()
                 for _ in range(nwords)]
        parts = []
        total = 0
        idx = rs.randint(0, nwords, size=n // 2 + 2)
        for i in idx:
            w = words[i] + b" "
            parts.append(w)
            total += len(w)
            if total >= n:
                break
        return np.frombuffer(b"".join(parts)[:n], dtype=np.uint8).copy()
    if kind == "lz":
        out = np.empty(n, dtype=np.uint8)
        pos = 0
        while pos < n:
            if pos == 0 or rs.randint(0, 3) == 0:
                ln = int(rs.randint(1, 40))
                ln = min(ln, n - pos)
                out[pos:pos + ln] = rs.randint(0, 256, size=ln)
                pos += ln
            else:
                off = int(rs.randint(1, min(pos, 70000) + 1))
                ln = int(rs.randint(4, 300)) if rs.randint(0, 8) else int(rs.randint(300, 5000))
                ln = min(ln, n - pos)
                for k in range(ln):  # overlapping copy, byte-serial
                    out[pos + k] = out[pos + k - off]
                pos += ln
        return out
    if kind in ("text_fox", "text_mtg"):
        c = corpus("fox" if kind == "text_fox" else "mtg")
        start = int(rs.randint(0, len(c)))
        reps = (start + n) // len(c) + 1
        return np.frombuffer((c * reps)[start:start + n], dtype=np.uint8).copy()
    raise KeyError(kind)


def tiled_block(source, k, block=65536):
    """Block k of the infinite tiling of `source`."""
    L = len(source)
    start = (k * block) % L
    reps = (start + block) // L + 1
    return (source * reps)[start:start + block]


def fnv1a64(b):
    h = 1469598103934665603
    for x in bytes(b):
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h
