#!/usr/bin/env python3
"""tools/pmc_r6_sum.py <gpurun_out/prof_r06> <profiles/> <commit> -- condenses the passes of tools/profile_r06.sh: per workload
the kernel-trace stats (profiles/r06_<w>_kernel_stats.txt), the per-launch medians of every counter for the workload's
dominant kernel (profiles/r06_<w>_pmc.txt) and the JSON bench.py reads (profiles/pmc_<w>_<blocks>.json: HBM bytes per
launch by MI355X_MICROARCH.md's rule -- FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x 2 on gfx950 --, LDS bank
conflicts per LDS-active cycle, waves resident per SIMD).  New in round 6: every JSON says which library was measured
(`library_source_id` = the `src` field of sz_version(), `library_version`) and at which commit (`collected_at_commit`,
handed in: the GPU box has no .git), so that bench.py can flag a file that is older than the library it runs."""
import collections, csv, glob, json, os, subprocess, sys
root, out = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else "?"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from pkg_loader import load_package
import re
version = load_package().lib().sz_version().decode()
src_id = (re.search(r"src ([0-9a-f]{16})", version) or [None, None])[1]
W = {"c3": ("compress_blocks_kernel", 32768, "bench.py --steps 8 --warmup 2 --no-cpu --no-extra"),
     "text": ("compress_blocks_team_kernel", 16384, "tools/quickbench.py text 16384 5"),
     "c5": ("compress_blocks_team_kernel", 16384, "tools/quickbench.py c5 16384 8"),
     "mix": ("compress_blocks_team_kernel", 16386, "tools/quickbench.py mix 16386 5")}
TILES = {"text": 16384 * 1010.0}
for w, (kern, nb, cmd) in W.items():
    stats_dir = os.path.join(root, w + "_stats")
    if os.path.isdir(stats_dir):
        txt = subprocess.check_output([sys.executable, os.path.join(out, "summarize.py"), stats_dir,
                                       "rocprofv3 --kernel-trace --stats -- python3 %s (round 6)" % cmd]).decode()
        open(os.path.join(out, "r06_%s_kernel_stats.txt" % w), "w").write(txt)
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(os.path.join(root, w + "_pmc*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("szk::", "")
                if k.startswith("compress_blocks"):
                    per_kernel[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    med = {k: {c: sorted(v)[len(v) // 2] for c, v in d.items()} for k, d in per_kernel.items()}
    n_of = {k: {c: len(v) for c, v in d.items()} for k, d in per_kernel.items()}
    if kern not in med:
        print("no counters for", w)
        continue
    c = med[kern]
    res = {"note": "%s, %d x 64 KiB blocks, %s (round 6); per-launch medians" % (w, nb, cmd), "kernel": "szk::" + kern,
           "library_version": version, "library_source_id": src_id, "collected_at_commit": commit,
           "dispatches": n_of[kern], "counters_per_launch": c}
    # a routed launch has two compress kernels at work: the other one's traffic belongs to the launch too
    other = [k for k in med if k != kern and med[k].get("FETCH_SIZE", 0) + med[k].get("WRITE_SIZE", 0) > 0.02 * (c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0))]
    rd = 2 * c.get("FETCH_SIZE", 0) * 1024 + sum(2 * med[k].get("FETCH_SIZE", 0) * 1024 for k in other)
    wr = c.get("WRITE_SIZE", 0) * 1024 + sum(med[k].get("WRITE_SIZE", 0) * 1024 for k in other)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        res.update({"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_traffic_bytes_per_launch": rd + wr,
                    "hbm_traffic_includes_kernels": [kern] + other,
                    "fetch_bytes_raw_per_launch": rd / 2,
                    "traffic_note": "the guide's rule (FETCH_SIZE x 2: exact for wide coalesced streaming reads on gfx950).  For the compressible "
                                    "workloads the reads are 64-byte gathers (candidate strings, table look-ups) that the x 2 over-counts -- the raw "
                                    "figure is fetch_bytes_raw_per_launch -- and FETCH_SIZE counts what the L2 fetches over the fabric, Infinity Cache "
                                    "hits included, not only HBM"})
    if c.get("SQ_LDS_IDX_ACTIVE"):
        res["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
    if "SQ_WAVE_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
        res["resident_waves_per_simd"] = c["SQ_WAVE_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0)
        res["wait_frac"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
    json.dump(res, open(os.path.join(out, "pmc_%s_%d.json" % (w, nb)), "w"), indent=1)
    lines = ["rocprofv3 --pmc, workload %s: %d blocks of 64 KiB per launch, %s; per-launch medians (round 6, tools/profile_r06.sh)." % (w, nb, cmd),
             "SQ_*_CYCLES / ACTIVE / WAIT: units of 4 clocks; GRBM_GUI_ACTIVE is summed over the 8 XCDs.", ""]
    for k in sorted(med):
        lines.append("%-34s" % k)
        for cn in sorted(med[k]):
            lines.append("  %-28s %14.4g   (n=%d)" % (cn, med[k][cn], n_of[k][cn]))
    o = c
    if "SQ_INSTS_VALU" in o:
        tot = sum(o.get(x, 0) for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
        lines.append("")
        if w in TILES:
            t = TILES[w]
            lines.append("%s: instructions per tile of 64 positions (1 010 tiles per block): VALU %.0f, SALU %.0f, LDS %.0f, VMEM %.0f, branch %.0f (sum without branches %.0f); SALU share %.0f %%" % (
                kern, o["SQ_INSTS_VALU"] / t, o["SQ_INSTS_SALU"] / t, o.get("SQ_INSTS_LDS", 0) / t,
                (o.get("SQ_INSTS_VMEM_RD", 0) + o.get("SQ_INSTS_VMEM_WR", 0)) / t, o.get("SQ_INSTS_BRANCH", 0) / t, tot / t, 100 * o["SQ_INSTS_SALU"] / tot))
        lines.append("%s: instructions per input byte: %.2f" % (kern, tot / (nb * 65536.0)))
    if "TCC_MISS_sum" in o and o.get("TCC_REQ_sum"):
        lines.append("L2 misses %.0f %% of %.3g requests; fabric reads %.3g (%.1f per 64 input bytes), writes %.3g" % (
            100 * o["TCC_MISS_sum"] / o["TCC_REQ_sum"], o["TCC_REQ_sum"], o.get("TCC_EA0_RDREQ_sum", 0),
            o.get("TCC_EA0_RDREQ_sum", 0) / (nb * 1024.0), o.get("TCC_EA0_WRREQ_sum", 0)))
    if "hbm_traffic_bytes_per_launch" in res:
        lines.append("HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB): read %.4g, written %.4g (kernels counted: %s)" % (rd, wr, ", ".join([kern] + other)))
    if "wait_frac" in res:
        lines.append("waves wait %.0f %% of their cycles; %.2f waves resident per SIMD; LDS bank conflicts %.0f %% of LDS-active cycles" % (
            100 * res["wait_frac"], res["resident_waves_per_simd"], 100 * res.get("lds_bank_conflict_frac", 0)))
    if "SQ_ACTIVE_INST_SCA" in o and o.get("GRBM_GUI_ACTIVE"):
        slots = o["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0
        lines.append("per SIMD issue slot (4 clocks): scalar port busy %.2f, vector ALU %.2f" % (o["SQ_ACTIVE_INST_SCA"] / slots, o.get("SQ_ACTIVE_INST_VALU", 0) / slots))
    open(os.path.join(out, "r06_%s_pmc.txt" % w), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-8:]))
