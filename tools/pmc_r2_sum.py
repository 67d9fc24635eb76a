#!/usr/bin/env python3
"""Sums the counter CSVs of tools/pmc_r2.sh per kernel and counter (average per dispatch of the compress kernels)."""
import csv, glob, os, sys
from collections import defaultdict
O = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(O, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if "compress_blocks" not in k:
                continue
            acc[k.split("(")[0]][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
for k, cs in acc.items():
    print("kernel", k)
    for c, vals in sorted(cs.items()):
        per = defaultdict(float)
        for d, v in vals:
            per[d] += v
        xs = sorted(per.values())
        xs = [x for x in xs if x > 0] or [0]
        print("  %-36s n=%d  max %.4g  median %.4g" % (c, len(xs), xs[-1], xs[len(xs) // 2]))
