#!/bin/bash
# Diagnostic: variant builds of kernels_wide.hip, each as its own library directory syllable_detector_swift_amd/lib_<name>/
# (run with SYLDET_LIB=.../lib_<name>/libsyldet.so):   tools/wide_variants.sh name "flags" [name "flags" ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/syllable_detector_swift_amd/csrc
make -C $CS -j6 >/dev/null
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  out=$ROOT/syllable_detector_swift_amd/lib_$name
  rm -rf $out && mkdir -p $out/obj && cp $ROOT/syllable_detector_swift_amd/lib/obj/*.o $out/obj/
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $flags -I$ROOT/include -c $CS/kernels_wide.hip -o $out/obj/kernels_wide.hip.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libsyldet.so $out/obj/*.o -ldl -lpthread
  echo built lib_$name
done
