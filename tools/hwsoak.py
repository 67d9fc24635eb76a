#!/usr/bin/env python3
"""tools/hwsoak.py [cases] [seed] -- SZ_FLAG_HW_COMPAT on random blocks against tests/hw_model.py (bytes) and the
software decoder (round trip)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzzgen, hw_model, oracle_lib
from gpu_util import compress_many
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
def varint(n):
    o = bytearray()
    while n >= 0x80:
        o.append((n & 0x7f) | 0x80); n >>= 7
    o.append(n); return bytes(o)
bad = 0; done = 0; t0 = time.time()
while done < cases:
    blocks = []
    for i in range(min(500, cases - done)):
        k = fuzzgen.KINDS[int(rs.randint(len(fuzzgen.KINDS)))]
        r = rs.randint(0, 10)
        n = int(rs.randint(0, 300)) if r < 3 else int(rs.randint(0, 5000)) if r < 7 else int(rs.randint(0, 65537))
        blocks.append(fuzzgen.gen(k, int(rs.randint(1 << 30)), n).tobytes())
    got = compress_many(blocks, varint=True, hw_compat=True)
    for b, g in zip(blocks, got):
        if g != varint(len(b)) + hw_model.compress_bytes(b) or oracle_lib.uncompress(g) != b:
            bad += 1
            print("MISMATCH len %d" % len(b), flush=True)
    done += len(blocks)
    print("%d cases, %d mismatches, %d s" % (done, bad, time.time() - t0), flush=True)
print("hwsoak done: %d cases, %d mismatches" % (done, bad))
sys.exit(1 if bad else 0)
