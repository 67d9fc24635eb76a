#!/usr/bin/env python3
"""tools/streamsoak.py -- soak of the host decompressor on multi-fragment streams: intact, spliced at odd
output offsets (valid, but not made of 64 KiB fragments), damaged.  Every verdict and every byte is compared
with the oracle.  usage: streamsoak.py <cases> <seed>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzzgen, oracle_lib
from pkg_loader import load_package
P = load_package()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
kinds = [k for k in fuzzgen.KINDS]

def varint(n):
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7f) | 0x80); n >>= 7
    out.append(n)
    return bytes(out)

def body_of(data):
    c = oracle_lib.raw_compress(data)
    k = 1
    while c[k - 1] & 0x80:
        k += 1
    return c[k:]

def piece():
    parts = []
    if rs.rand() < 0.02:  # now and then a stream of several index chunks (64 segments of 16 KiB each)
        k = kinds[int(rs.randint(len(kinds)))]
        return fuzzgen.gen(k, int(rs.randint(1 << 30)), int(rs.randint(40, 200)) * 65536 + int(rs.randint(0, 65536))).tobytes()
    for _ in range(int(rs.randint(1, 6))):
        k = kinds[int(rs.randint(len(kinds)))]
        parts.append(fuzzgen.gen(k, int(rs.randint(1 << 30)), int(rs.randint(1, 5)) * 65536 if rs.rand() < 0.5 else int(rs.randint(1000, 300000))).tobytes())
    return b"".join(parts)

t0 = time.time(); bad = 0; paths = {"ok": 0, "err": 0}
for i in range(cases):
    os.environ["SZ_TEST_SEG_SHIFT"] = str(12 + (i // 5) % 5)  # every segment size of the fragment index in turn
    mode = i % 5
    a = piece()
    if mode == 0:      # intact, fragment-aligned
        data = a; s = oracle_lib.raw_compress(a)
    elif mode == 1:    # two bodies spliced at an odd output offset: valid, not fragment-aligned
        b = piece(); data = a + b; s = varint(len(data)) + body_of(a) + body_of(b)
    else:              # damaged
        s = bytearray(oracle_lib.raw_compress(a))
        if mode == 2:
            for _ in range(int(rs.randint(1, 4))):
                s[int(rs.randint(0, len(s)))] ^= 1 << int(rs.randint(0, 8))
        elif mode == 3:
            s = s[:int(rs.randint(1, len(s)))]
        else:
            pos = int(rs.randint(1, len(s))); s[pos:pos] = bytes(rs.randint(0, 256, size=int(rs.randint(1, 5)), dtype=np.uint8))
        s = bytes(s); data = None
    try:
        want = ("ok", oracle_lib.uncompress(s))
    except ValueError:
        want = ("err", None)
    try:
        got = ("ok", P.uncompress(s))
    except (P.SnappyHipError, Exception) as e:  # noqa
        got = ("err", None)
    if got != want or (data is not None and (want[0] != "ok" or want[1] != data)):
        bad += 1
        print("MISMATCH case %d mode %d len %d: got %s want %s" % (i, mode, len(s), got[0], want[0]), flush=True)
    paths[got[0]] += 1
    if (i + 1) % 100 == 0:
        print("%d cases, %d mismatches, %d s, verdicts %s" % (i + 1, bad, time.time() - t0, paths), flush=True)
print("streamsoak done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
