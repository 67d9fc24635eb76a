# rocprofv3 --pmc passes for the L2 (TCC) behaviour of one workload; output: gpurun_out/pmc_l2_<tag>/
# usage: bash tools/pmc_l2.sh <tag> <workload> <blocks>   (env SZ_KERNEL respected)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; W=$2; NB=$3
O=$R/gpurun_out/pmc_l2_$TAG
rm -rf $O; mkdir -p $O
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_READ_sum TCC_WRITE_sum TCC_WRITEBACK_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -d $O/p$i -o p$i --output-format csv -- python3 $R/tools/quickbench.py $W $NB 3 > $O/log$i.txt 2>&1 || { echo "pass $i failed"; tail -5 $O/log$i.txt; }
  echo "pass $i done"
done
python3 $R/tools/pmc_r2_sum.py $O > $O/summary.txt 2>&1; cat $O/summary.txt
