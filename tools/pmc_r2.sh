# rocprofv3 --pmc passes over one workload (tools/quickbench.py), counters per pass; output: gpurun_out/pmc_r2_<tag>/
# usage: bash tools/pmc_r2.sh <tag> <workload> <blocks>   (env SZ_KERNEL respected)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; W=$2; NB=$3
O=$R/gpurun_out/pmc_r2_$TAG
rm -rf $O; mkdir -p $O
i=0
for set in "GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -d $O/p$i -o p$i --output-format csv -- python3 $R/tools/quickbench.py $W $NB 3 > $O/log$i.txt 2>&1 || { echo "pass $i failed"; tail -5 $O/log$i.txt; }
  echo "pass $i done"
done
python3 $R/tools/pmc_r2_sum.py $O > $O/summary.txt 2>&1; cat $O/summary.txt
