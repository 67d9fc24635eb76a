#!/bin/bash
# tools/hostcmp.sh [lib.so] -- call-to-call times of the host-buffer API on 2 GiB of incompressible data, C++ caller
# (tools/hostpipe.cc; a -DSZ_HOST_TIMING build of the library -- tools/mkvariant.sh hosttime -DSZ_HOST_TIMING -- names
# the phases on stderr), then the same through Python in a process that has torch's streams in it (tools/hostapi_bench.py).
# Round 6: one call in three took 75 ms instead of 47 (EXPERIMENTS.md R6.9).
set -e
cd $GRAFT_REPO_ROOT
LIB=${1:-compression-accelerator_amd/csrc/libsnappy_hip.so}
mkdir -p gpurun_out/hostcmp
python3 - <<'PY'
import sys; sys.path.insert(0,'tests')
import fuzzgen
b = fuzzgen.java_random_ascii(100000, 4444)
open('/tmp/c3.bin','wb').write(b if isinstance(b, (bytes, bytearray)) else b.tobytes())
PY
test -s /tmp/c3.bin
g++ -O2 -std=c++11 -o /tmp/hostpipe tools/hostpipe.cc -ldl
timeout -k 5 120 /tmp/hostpipe $LIB /tmp/c3.bin 2048 12 > gpurun_out/hostcmp/hostpipe_c3.log 2>&1
grep -E "compress   call|string     call|round trip" gpurun_out/hostcmp/hostpipe_c3.log
timeout -k 5 300 python3 tools/hostapi_bench.py 2 > gpurun_out/hostcmp/hostapi_bench.json 2> gpurun_out/hostcmp/hostapi_bench.err
grep -E "compress_ms|compress_GBps|bytes_equal|round_trip" gpurun_out/hostcmp/hostapi_bench.json
