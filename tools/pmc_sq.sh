# rocprofv3 --pmc passes: instruction fetch, LDS and vector-memory queue counters of one workload; output: gpurun_out/pmc_sq_<tag>/
# usage: bash tools/pmc_sq.sh <tag> <workload> <blocks>   (env SZ_KERNEL respected)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; W=$2; NB=$3
O=$R/gpurun_out/pmc_sq_$TAG
rm -rf $O; mkdir -p $O
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_CYCLES SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -d $O/p$i -o p$i --output-format csv -- python3 $R/tools/quickbench.py $W $NB 3 > $O/log$i.txt 2>&1 || { echo "pass $i failed"; tail -5 $O/log$i.txt; }
  echo "pass $i done"
done
python3 $R/tools/pmc_r2_sum.py $O > $O/summary.txt 2>&1; cat $O/summary.txt
