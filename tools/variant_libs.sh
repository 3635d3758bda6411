#!/bin/bash
# Diagnostic: variant builds of ONE source file of the library, each as its own library directory
# syllable_detector_swift_amd/lib_<name>/ (run with SYLDET_LIB=.../lib_<name>/libsyldet.so, e.g. through tools/ab_env.py
# "name:SYLDET_LIB=..."):      tools/variant_libs.sh kernels_wide.hip name "flags" [name "flags" ...]
# (the file's own extra flags in csrc/Makefile are NOT applied: pass them if the variant should keep them)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/syllable_detector_swift_amd/csrc
make -C $CS -j6 >/dev/null
src=$1; shift
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  out=$ROOT/syllable_detector_swift_amd/lib_$name
  rm -rf $out && mkdir -p $out/obj && cp $ROOT/syllable_detector_swift_amd/lib/obj/*.o $out/obj/
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $flags -I$ROOT/include -c $CS/$src -o $out/obj/$src.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libsyldet.so $out/obj/*.o -ldl -lpthread
  echo built lib_$name
done
