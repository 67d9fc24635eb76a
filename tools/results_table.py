#!/usr/bin/env python3
"""tools/results_table.py -- the reference's reporting for this path.

Runs the snappy::-facade harness (drivers/sw_benchmark.cc, the shape of the reference's
workloads/snappy/overlay/root/compress.cc: 20 runs per input) over the 36 checked-in
benchmark-data inputs and prints
  * the CSV of benchmark/sw_results.csv:1 -- type;length;<time>;compressed-length -- with the mean
    nanoseconds of one snappy::Compress call (host buffers, PCIe included) in place of Rocket cycles;
  * the ratio / efficiency table benchmark/csv_scan.py:19-21 derives from it
    (ratio = length / compressed-length, efficiency = time / length), next to the compressed
    lengths the reference published.
usage (GPU box): python tools/results_table.py > profiles/rNN_sw_results.txt
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import build_util  # noqa: E402
import fuzzgen  # noqa: E402


def main():
    d = tempfile.mkdtemp()
    exe = build_util.cxx(os.path.join(build_util.DRIVERS, "sw_benchmark.cc"), os.path.join(d, "sw_benchmark"))
    with open(os.path.join(fuzzgen.GOLDEN_DIR, "sw_results_sizes.json")) as f:
        pins = json.load(f)
    ref = {(r["type"], r["length"]): r["compressed_length"] for r in pins["rows"]}
    sizes = (10, 20, 50, 100, 200, 500, 1000, 2000, 5000, 10000, 20000, 50000)
    rows = []
    print("type;length;ns;compressed-length")
    for typ, prefix in pins["file_prefix"].items():
        files = [os.path.join(fuzzgen.DATA_DIR, "benchmark-data", "%s-%d.txt" % (prefix, n)) for n in sizes]
        out = subprocess.run([exe, typ] + files, capture_output=True, text=True, timeout=600, check=True).stdout.split()
        for line in out[1:]:
            print(line)
            t, length, ns, clen = line.split(";")
            rows.append((t, int(length), int(ns), int(clen)))
    print()
    print("%-8s %8s %10s %14s %12s %12s" % ("type", "length", "ratio", "ns per byte", "compressed", "reference"))
    for t, length, ns, clen in rows:
        print("%-8s %8d %10.4f %14.3f %12d %12d" % (t, length, length / clen, ns / length, clen, ref[(t, length)]))
    same = sum(1 for t, length, ns, clen in rows if clen == ref[(t, length)])
    print("\ncompressed-length equal to the reference's benchmark/sw_results.csv in %d of %d rows" % (same, len(rows)))


if __name__ == "__main__":
    main()
