# one rocprofv3 --pmc pass: instruction counts by class of one workload for a given build (SZ_AB_LIB); output: gpurun_out/pmc_insts_<tag>/
# usage: bash tools/pmc_insts.sh <tag> <workload> <blocks>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; W=$2; NB=$3
O=$R/gpurun_out/pmc_insts_$TAG
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_BRANCH -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/quickbench.py $W $NB 3 > $O/log1.txt 2>&1 || { echo "pass failed"; tail -5 $O/log1.txt; }
python3 $R/tools/pmc_r2_sum.py $O > $O/summary.txt 2>&1; grep -A12 dual_kernel $O/summary.txt
