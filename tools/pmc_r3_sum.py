#!/usr/bin/env python3
"""tools/pmc_r3_sum.py <gpurun_out/prof_r03> -- condenses the text passes of tools/profile_r03.sh (team kernel and
multi-wave kernel, tools/quickbench.py text 4096) into profiles/r03_text_pmc.txt: per-launch medians per counter."""
import collections, csv, glob, os, sys
root = sys.argv[1]
out = {}
for k in ("team", "dual"):
    agg = collections.defaultdict(list)
    for d in glob.glob(os.path.join(root, "text_%s_*" % k)):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                kn = r["Kernel_Name"]
                if ("team_kernel" in kn and k == "team") or ("dual_kernel" in kn and k == "dual"):
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[k] = {c: sorted(v)[len(v) // 2] for c, v in agg.items() if v}
names = sorted(set(out["team"]) | set(out["dual"]))
print("rocprofv3 --pmc, text workload (all-mtg-cards tiled), 4096 blocks of 64 KiB = 268 MB per launch; per-launch medians.")
print("team = szk::compress_blocks_team_kernel (round 3 default for compressible launches: per workgroup a 2-wave team on the")
print("LDS table + 2 waves with tables in global memory), dual = szk::compress_blocks_dual_kernel (rounds 1-2: 3 independent waves")
print("per workgroup, 2 of them with tables in global memory).  1 010 tiles per block.  SQ_*_CYCLES / ACTIVE / WAIT: units of 4 clocks.\n")
print("%-28s %14s %14s %8s" % ("counter", "team", "dual", "ratio"))
for c in names:
    a, b = out["team"].get(c), out["dual"].get(c)
    print("%-28s %14s %14s %8s" % (c, "%.4g" % a if a is not None else "-", "%.4g" % b if b is not None else "-",
                                   "%.2f" % (a / b) if a and b else "-"))
tiles = 4096 * 1010.0
for k in ("team", "dual"):
    o = out[k]
    if "SQ_INSTS_VALU" in o:
        tot = sum(o.get(c, 0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
        print("\n%s: instructions per tile: VALU %.0f, SALU %.0f, LDS %.0f, VMEM %.0f, branch %.0f (sum without branches %.0f)" % (
            k, o["SQ_INSTS_VALU"] / tiles, o["SQ_INSTS_SALU"] / tiles, o.get("SQ_INSTS_LDS", 0) / tiles,
            (o.get("SQ_INSTS_VMEM_RD", 0) + o.get("SQ_INSTS_VMEM_WR", 0)) / tiles, o.get("SQ_INSTS_BRANCH", 0) / tiles, tot / tiles))
    if "TCC_MISS_sum" in o and "TCC_REQ_sum" in o:
        print("%s: L2 misses %.0f %% of %.3g requests; fabric reads %.3g (%.1f per 64 input bytes), writes %.3g" % (
            k, 100 * o["TCC_MISS_sum"] / o["TCC_REQ_sum"], o["TCC_REQ_sum"], o.get("TCC_EA0_RDREQ_sum", 0),
            o.get("TCC_EA0_RDREQ_sum", 0) / (4096 * 1024.0), o.get("TCC_EA0_WRREQ_sum", 0)))
    if "SQ_WAIT_ANY" in o and "SQ_WAVE_CYCLES" in o:
        print("%s: waves wait %.0f %% of their cycles" % (k, 100 * o["SQ_WAIT_ANY"] / o["SQ_WAVE_CYCLES"]))
    if "SQ_LDS_BANK_CONFLICT" in o and o.get("SQ_LDS_IDX_ACTIVE"):
        print("%s: LDS bank conflicts %.0f %% of LDS-active cycles" % (k, 100 * o["SQ_LDS_BANK_CONFLICT"] / o["SQ_LDS_IDX_ACTIVE"]))
