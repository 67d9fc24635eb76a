# Round-6 evidence under profiles/ (run on the GPU box through gpurun): for each of the four workloads of the bench
# line -- C3 (the bench command itself), text, low-entropy C5, C4 mix (tools/quickbench.py at the bench's sizes) -- a
# kernel-trace stats pass and PMC passes (HBM bytes; SQ occupancy / waits / LDS bank conflicts; instruction mix;
# L2 and fabric requests), one counter group per pass, the program directly behind `--`.
# usage (through gpurun): bash tools/profile_r06.sh <commit the tree was pushed from>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r06
rm -rf $O; mkdir -p $O
SETS=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS")
run() {  # <tag> <program and arguments...>
  tag=$1; shift
  timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o s -- "$@" > $O/${tag}_stats.log 2>&1 || echo "$tag stats failed"
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout -k 10 250 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${tag}_pmc$i -o p -- "$@" > $O/${tag}_pmc$i.log 2>&1 || echo "$tag pmc $i failed"
  done
  echo "$tag done"
}
run c3 python3 $R/bench.py --steps 8 --warmup 2 --no-cpu --no-extra
run text python3 $R/tools/quickbench.py text 16384 5
run c5 python3 $R/tools/quickbench.py c5 16384 8
run mix python3 $R/tools/quickbench.py mix 16386 5
python3 $R/tools/pmc_r6_sum.py $O $R/profiles "${1:-?}"
