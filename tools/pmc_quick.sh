#!/bin/bash
# tools/pmc_quick.sh <tag> <workload> <blocks> "<counters...>" ["<counters...>" ...] -- rocprofv3 --pmc passes over tools/quickbench.py
# (env SZ_AB_LIB = another build, SZ_KERNEL as usual); one pass per counter set; per-kernel medians printed.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; wl=$2; nb=$3; shift 3
O=$R/gpurun_out/pmcq_$tag
rm -rf $O; mkdir -p $O
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python3 $R/tools/quickbench.py $wl $nb 5 > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$O" <<'PY'
import collections, csv, glob, os, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("szk::compress"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    for c, v in sorted(d.items()):
        v.sort()
        print("%-40s %-26s median %.4g (n=%d)" % (k, c, v[len(v) // 2], len(v)))
PY
