#!/usr/bin/env python3
"""tools/hostrate.py [MiB] -- PCIe-inclusive rate of the host-buffer API through the snappy:: facade, on pageable host
memory as a caller has it: builds drivers/sw_benchmark.cc and runs it on a tiled corpus file, 20 calls averaged (the
reference's own loop, workloads/snappy/overlay/root/compress.cc:26-34).  Two call shapes per input:
  string  snappy::Compress(input, n, &std::string)          compress.cc:17
  raw     snappy::RawCompress(input, n, buffer, &length)    tests/swcompression.cc:45
and the round trip through snappy::Uncompress (timed once, after them).  Prints one JSON object at the end."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import build_util, fuzzgen
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = tempfile.mkdtemp()
exe = build_util.cxx(os.path.join(build_util.DRIVERS, "sw_benchmark.cc"), os.path.join(d, "sw_benchmark"))
res = {"MiB": mib}
for name, corpus in (("text", fuzzgen.corpus("mtg")), ("random", fuzzgen.corpus("random_ascii"))):
    path = os.path.join(d, name + ".bin")
    with open(path, "wb") as f:
        total, piece = mib << 20, corpus * max(1, (64 << 20) // len(corpus))
        while total > 0:  # (the tiling of the corpus, written piecewise: no 2 GiB Python object)
            f.write(piece[:total])
            total -= min(total, len(piece))
    for shape, flag in (("string", []), ("raw", ["-r"])):
        r = subprocess.run([exe] + flag + [name, path], capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print(name, shape, "FAILED", r.stderr[-500:])
            sys.exit(1)
        t, length, ns, clen = r.stdout.split()[1].split(";")
        rate = int(length) / int(ns)
        res["%s_%s_GBps" % (name, shape)] = round(rate, 2)
        print("%s, %s: %d MiB -> %s bytes, %.1f ms per call = %.2f GB/s (copy in + kernels + pack + copy out, pageable host memory; round trip ok)"
              % (name, shape, mib, clen, int(ns) / 1e6, rate), flush=True)
        for ln in r.stderr.splitlines():
            if ln.startswith("uncompress;") and shape == "string":
                _, length, uns = ln.split(";")
                res["%s_uncompress_GBps" % name] = round(int(length) / int(uns), 2)
                print("%s: snappy::Uncompress %.1f ms per call = %.2f GB/s of output" % (name, int(uns) / 1e6, int(length) / int(uns)), flush=True)
    os.unlink(path)
print(json.dumps(res))
