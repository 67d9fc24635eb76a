cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_decomp
rm -rf $O; mkdir -p $O
for k in text c3; do
  timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$k -o $k -- python3 $R/tools/decompbench.py $k 16384 > $O/$k.log 2>&1 || echo "$k failed"
  echo "$k done"
done
