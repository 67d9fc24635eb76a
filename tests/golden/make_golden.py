#!/usr/bin/env python3
"""Generates tests/golden/*.json and copies the data fixtures.

Run in the BUILD CONTAINER only (it needs /root/reference for the data files and
libsnappy 1.1.8 for the expected outputs):

    python tests/golden/make_golden.py

Expected outputs come from google/snappy 1.1.8 -- /opt/conda/lib/libsnappy.so.1.1.8, called
through its C API (snappy-c.h: snappy_compress).  That is the library version the reference's
software path (tests/swcompression.cc:45, workloads/snappy/overlay/root/compress.cc:17) was
built against as far as can be established; the reference's own
google/snappy submodule is empty.  pyarrow's bundled snappy is a newer release with different
output bytes and is deliberately NOT used.

What is written:
  data/                       data fixtures copied verbatim from the reference's data/ and
                              benchmark/benchmark-data/ directories (inputs only; data, not code)
  golden_files.json           per-file: input sha256, compressed length, sha256 + fnv1a64 of the
                              compressed stream, full compressed hex for inputs <= 8 KiB,
                              per-64KiB-block RawCompress lengths/sha256 for the big corpora
  golden_fuzz.json            (kind, seed, n) -> length + sha256 of RawCompress output for the
                              deterministic generators in tests/fuzzgen.py
  hw_results_sizes.json       the compressed-length column of the reference's benchmark/hw_results.csv
                              (the accelerator's own sizes: SURVEY 8f-4, tests/hw_model.py)
  sw_results_sizes.json       the compressed-length column of the reference's
                              benchmark/sw_results.csv (the only numbers the reference itself
                              pins for this path)
"""
import ctypes
import hashlib
import json
import lzma
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import fuzzgen  # noqa: E402

REF = "/root/reference"
LIBSNAPPY = "/opt/conda/lib/libsnappy.so.1.1.8"
BLOCK = 65536


def load_snappy():
    lib = ctypes.CDLL(LIBSNAPPY)
    lib.snappy_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                    ctypes.POINTER(ctypes.c_size_t)]
    lib.snappy_compress.restype = ctypes.c_int
    lib.snappy_max_compressed_length.argtypes = [ctypes.c_size_t]
    lib.snappy_max_compressed_length.restype = ctypes.c_size_t
    return lib


_lib = None


def snappy_compress(data):
    global _lib
    if _lib is None:
        _lib = load_snappy()
    data = bytes(data)
    cap = _lib.snappy_max_compressed_length(len(data))
    buf = ctypes.create_string_buffer(cap)
    n = ctypes.c_size_t(cap)
    rc = _lib.snappy_compress(data, len(data), buf, ctypes.byref(n))
    assert rc == 0
    return buf.raw[:n.value]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def file_entry(name, data, want_blocks=False, hex_limit=8192):
    out = snappy_compress(data)
    e = {"name": name, "in_len": len(data), "in_sha256": sha(data), "out_len": len(out),
         "out_sha256": sha(out), "out_fnv1a64": "%016x" % fuzzgen.fnv1a64(out)}
    if len(data) <= hex_limit:
        e["out_hex"] = out.hex()
    if want_blocks:
        blocks = []
        for off in range(0, len(data), BLOCK):
            o = snappy_compress(data[off:off + BLOCK])
            blocks.append({"in_len": len(data[off:off + BLOCK]), "out_len": len(o), "out_sha256": sha(o)})
        e["blocks"] = blocks
    return e


def main():
    data_dir = os.path.join(HERE, "data")
    os.makedirs(os.path.join(data_dir, "benchmark-data"), exist_ok=True)

    # ---- data fixtures
    for f in ("fox-in-socks.txt", "randomASCII.txt", "alignerTestData.txt"):
        shutil.copyfile(os.path.join(REF, "data", f), os.path.join(data_dir, f))
    with open(os.path.join(REF, "data", "all-mtg-cards.txt"), "rb") as f:
        mtg = f.read()
    xz_path = os.path.join(data_dir, "all-mtg-cards.txt.xz")
    if not os.path.exists(xz_path):
        with lzma.open(xz_path, "wb", preset=9) as f:
            f.write(mtg)
    bench_names = []
    for kind in ("random", "real", "repeating"):
        for n in (10, 20, 50, 100, 200, 500, 1000, 2000, 5000, 10000, 20000, 50000):
            nm = "%s-%d.txt" % (kind, n)
            shutil.copyfile(os.path.join(REF, "benchmark", "benchmark-data", nm),
                            os.path.join(data_dir, "benchmark-data", nm))
            bench_names.append(nm)

    # ---- sizes pinned by the reference itself
    rows = []
    with open(os.path.join(REF, "benchmark", "sw_results.csv")) as f:
        header = f.readline().strip()
        assert header == "type;length;cycles;compressed-length", header
        for line in f:
            t, length, _cycles, clen = line.strip().split(";")
            rows.append({"type": t, "length": int(length), "compressed_length": int(clen)})
    with open(os.path.join(HERE, "sw_results_sizes.json"), "w") as f:
        json.dump({"source": "reference benchmark/sw_results.csv (type;length;cycles;compressed-length)",
                   "file_prefix": {"random": "random", "real": "real", "repeat": "repeating"},
                   "rows": rows}, f, indent=1)

    # ---- the accelerator's own sizes (tests/hw_model.py, tests/test_hw_model.py)
    rows = []
    with open(os.path.join(REF, "benchmark", "hw_results.csv")) as f:
        header = f.readline().strip()
        assert header == "type;length;cycles;compressed-length", header
        for line in f:
            t, length, _cycles, clen = line.strip().split(";")
            rows.append({"type": t, "length": int(length), "compressed_length": int(clen)})
    with open(os.path.join(HERE, "hw_results_sizes.json"), "w") as f:
        json.dump({"source": "reference benchmark/hw_results.csv (type;length;cycles;compressed-length): the accelerator's own output sizes, 512-entry table",
                   "file_prefix": {"random": "random", "real": "real", "repeat": "repeating"},
                   "rows": rows}, f, indent=1)

    # ---- file goldens
    thistle = (b"If Tid the thistle sifter lifts his sieve of sifted thistles, then he'll shift "
               b"the thistles in his sifting sieve to live unsifted.")
    assert len(thistle) == 130
    files = [file_entry("thistle", thistle)]
    fox = fuzzgen.corpus("fox")
    rnd = fuzzgen.corpus("random_ascii")
    files.append(file_entry("fox-in-socks.txt", fox))
    files.append(file_entry("randomASCII.txt", rnd, want_blocks=True))
    files.append(file_entry("all-mtg-cards.txt", mtg, want_blocks=True))
    for nm in bench_names:
        with open(os.path.join(data_dir, "benchmark-data", nm), "rb") as f:
            files.append(file_entry("benchmark-data/" + nm, f.read(), hex_limit=2048))
    # config C5 candidates and C3 windows
    files.append(file_entry("mtg[0:4096]x16", mtg[:4096] * 16))
    files.append(file_entry("fox[0:4096]x16", fox[:4096] * 16))
    files.append(file_entry("randomASCII[0:4096]x16", rnd[:4096] * 16))
    for k in (0, 1, 2, 7, 100, 16383):
        files.append(file_entry("randomASCII tiled block %d" % k, fuzzgen.tiled_block(rnd, k)))
    with open(os.path.join(HERE, "golden_files.json"), "w") as f:
        json.dump({"oracle": "libsnappy 1.1.8 snappy_compress", "files": files}, f, indent=1)

    # ---- fuzz goldens
    fuzz = []
    sizes_small = [0, 1, 2, 3, 4, 5, 13, 14, 15, 16, 17, 18, 19, 20, 31, 32, 33, 59, 60, 61, 62, 63, 64,
                   65, 66, 67, 68, 69, 255, 256, 257, 258, 1000, 2047, 2048, 2049, 4096, 5000]
    seed = 1
    for kind in fuzzgen.KINDS:
        for n in sizes_small:
            fuzz.append((kind, seed, n))
            seed += 1
        for n in (8191, 8192, 8193, 16384, 16385, 30000, 65521, 65535, 65536):
            fuzz.append((kind, seed, n))
            seed += 1
        for n in (65537, 70000, 131072, 200001):  # multi-fragment streams
            fuzz.append((kind, seed, n))
            seed += 1
    import numpy as np
    rs = np.random.RandomState(20261003)
    for _ in range(600):
        kind = fuzzgen.KINDS[int(rs.randint(0, len(fuzzgen.KINDS)))]
        n = int(rs.randint(0, 9000)) if rs.randint(0, 4) else int(rs.randint(9000, 65537))
        fuzz.append((kind, seed, n))
        seed += 1
    entries = []
    for kind, sd, n in fuzz:
        data = fuzzgen.gen(kind, sd, n).tobytes()
        out = snappy_compress(data)
        entries.append({"kind": kind, "seed": sd, "n": n, "out_len": len(out), "out_sha256": sha(out)})
    with open(os.path.join(HERE, "golden_fuzz.json"), "w") as f:
        json.dump({"oracle": "libsnappy 1.1.8 snappy_compress", "cases": entries}, f, indent=0)
    print("files:", len(files), "fuzz:", len(entries))


if __name__ == "__main__":
    main()
