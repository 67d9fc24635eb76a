#!/usr/bin/env python3
"""Condenses rocprofv3 --pmc passes (one directory per pass, csv) for szk::compress_blocks_kernel
into a small JSON + text summary committed under profiles/.

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced (16 B/lane) streaming read, so it is
doubled; WRITE_SIZE is exact for 16-B/lane streaming stores.  FETCH and WRITE need separate passes
(TCC slots), SQ/GRBM counters a third.

    python profiles/pmc_summary.py OUT.json "workload note" dir_fetch dir_write dir_sq
"""
import collections
import csv
import glob
import json
import os
import sys


def collect(d):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "compress_blocks_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    out, note = sys.argv[1], sys.argv[2]
    c, n = {}, {}
    for d in sys.argv[3:]:
        a, b = collect(d)
        c.update(a)
        n.update(b)
    res = {"note": note, "kernel": "szk::compress_blocks_kernel", "dispatches_averaged": n, "counters_per_launch": c}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        res["hbm_read_bytes_per_launch"] = 2 * c["FETCH_SIZE"] * 1024  # gfx950 correction: x2
        res["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
        res["hbm_traffic_bytes_per_launch"] = res["hbm_read_bytes_per_launch"] + res["hbm_write_bytes_per_launch"]
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        res["lds_bank_conflict_frac_of_lds_active"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        res["lds_bank_conflict_frac"] = res["lds_bank_conflict_frac_of_lds_active"]
    if "SQ_WAVE_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
        # waves resident per SIMD, averaged over the kernel: wave-residency (quad-cycles, summed over waves) over
        # the kernel's duration (GRBM_GUI_ACTIVE is summed over the 8 XCDs, in clocks) x 1 024 SIMDs
        res["resident_waves_per_simd"] = c["SQ_WAVE_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0)
    if "SQ_WAVE_CYCLES" in c and c.get("SQ_WAVES"):
        res["wave_cycles_per_wave_(quad_cycles)"] = c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"]
        res["wait_frac"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
        res["active_inst_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
