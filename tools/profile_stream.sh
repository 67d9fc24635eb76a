cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_stream
rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -o s -- python3 $R/tools/streamdecomp.py 1024 > $O/s.log 2>&1 || echo failed
echo done
