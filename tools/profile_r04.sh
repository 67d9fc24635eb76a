# Round-4 evidence under profiles/ (run on the GPU box through gpurun): kernel-trace stats of the bench command and
# of the text workload (team kernel), PMC passes for C3 (HBM traffic, LDS, occupancy; one counter group per pass) and
# for text (instruction mix, L2), and the bench line itself.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
rm -rf $O; mkdir -p $O
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o c3 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-extra > $O/c3.log 2>&1 || echo "c3 stats failed"
echo "c3 stats done"
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/text -o text -- python3 $R/tools/quickbench.py text 16384 5 > $O/text.log 2>&1 || echo "text stats failed"
echo "text stats done"
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -o p -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra > $O/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
  echo "pmc $tag done"
done
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/text_team_$tag -o p -- python3 $R/tools/quickbench.py text 4096 5 > $O/text_team_$tag.log 2>&1 || echo "pmc text $tag failed"
  echo "pmc text $tag done"
done
timeout -k 10 600 python3 $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 600 $O/bench.json
