#!/bin/bash
# tools/mkvariant.sh <tag> [-DFLAG ...] -- an alternative build of the library for A/B runs on the GPU box
# (tools/abmulti.py ab/libsnappy_hip_<tag>.so,...).  ab/ is git-ignored but travels with gpurun.
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/compression-accelerator_amd/csrc
out=$root/ab; mkdir -p $out/obj_$tag
for f in snappy_compress snappy_pack snappy_uncompress snappy_stream_index snappy_hwcompat; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -I$root/include -I$src "$@" -c $src/$f.hip -o $out/obj_$tag/$f.o &
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -I$root/include -I$src "$@" -c $src/snappy_abi.cpp -o $out/obj_$tag/snappy_abi.o &
wait
for f in snappy_compress snappy_pack snappy_uncompress snappy_stream_index snappy_hwcompat snappy_abi; do test -s $out/obj_$tag/$f.o || { echo "compile of $f failed"; rm -f $out/libsnappy_hip_$tag.so; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libsnappy_hip_$tag.so $out/obj_$tag/*.o -Wl,--version-script=$src/exports.map
rm -rf $out/obj_$tag
echo built $out/libsnappy_hip_$tag.so
