#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --stats (csv) output directory into a small text summary
that is committed under profiles/ (the raw directory under gpurun_out/ is scratch).

    python profiles/summarize.py gpurun_out/prof_dir "command line that was profiled" > profiles/rNN_name.txt
"""
import csv
import glob
import os
import sys


def main():
    d, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    print("# rocprofv3 --kernel-trace --stats summary")
    print("# command: %s" % cmd)
    for path in sorted(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)):
        print("# source: %s" % os.path.relpath(path))
        print("%-60s %8s %14s %14s %14s %8s" % ("kernel", "calls", "avg_ns", "min_ns", "max_ns", "pct"))
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Name"]
                if len(name) > 58:
                    name = name[:55] + "..."
                print("%-60s %8s %14.0f %14s %14s %8s" % (name, row["Calls"], float(row["AverageNs"]),
                                                       row["MinNs"], row["MaxNs"], row["Percentage"]))
    for path in sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)):
        with open(path) as f:
            rows = [r for r in csv.DictReader(f) if r["Kernel_Name"].startswith("szk::")]
        if not rows:
            continue
        print("\n# per-dispatch (szk:: kernels) from %s" % os.path.relpath(path))
        print("%-40s %10s %8s %6s %6s %6s %12s" % ("kernel", "grid_x", "wg_x", "vgpr", "sgpr", "lds", "duration_ns"))
        for r in rows:
            print("%-40s %10s %8s %6s %6s %6s %12d" % (
                r["Kernel_Name"].split("(")[0], r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"],
                r["SGPR_Count"], r["LDS_Block_Size"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))


if __name__ == "__main__":
    main()
