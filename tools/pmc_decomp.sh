cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_decomp
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "TA_BUSY_avr TA_BUSY_max TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -d $O/p$i -o p$i --output-format csv -- python3 $R/tools/decompbench.py text 16384 > $O/log$i.txt 2>&1 || { echo "pass $i failed"; tail -5 $O/log$i.txt; }
  echo "pass $i done"
done
