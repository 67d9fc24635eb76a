# HBM traffic of one workload (FETCH_SIZE / WRITE_SIZE, one pass each; units as in MI355X_MICROARCH.md); output gpurun_out/pmc_traffic_<tag>/
# usage: bash tools/pmc_traffic.sh <tag> <workload> <blocks>   (env SZ_AB_LIB, SZ_KERNEL respected)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; W=$2; NB=$3
O=$R/gpurun_out/pmc_traffic_$TAG
rm -rf $O; mkdir -p $O
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do  # (TCC_* groups time out under rocprofv3 on this pool)
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -d $O/p$i -o p$i --output-format csv -- python3 $R/tools/quickbench.py $W $NB 3 > $O/log$i.txt 2>&1 || { echo "pass $i failed"; tail -3 $O/log$i.txt; }
done
python3 - <<PY
import csv, glob, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, cs in acc.items():
    if "szk::" not in k: continue
    for c, per in sorted(cs.items()):
        xs = sorted(per.values()); print("%-40s %-22s n=%d median %.5g" % (k[:40], c, len(xs), xs[len(xs)//2]))
PY
