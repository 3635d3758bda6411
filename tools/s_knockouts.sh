#!/bin/bash
# Diagnostic: variant builds of the symmetric-fold kernel with one piece knocked out (results wrong by construction), each as
# its own library directory syllable_detector_swift_amd/lib_<name>/ for tools/ab_kernel.py:
#     tools/s_knockouts.sh && gpurun -- python tools/ab_kernel.py lib lib_nomax lib_nodft
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/syllable_detector_swift_amd/csrc
make -C $CS -j6 >/dev/null
for v in "$@"; do
  name=$(echo $v | tr 'A-Z' 'a-z' | sed 's/syldet_s_//')
  out=$ROOT/syllable_detector_swift_amd/lib_$name
  rm -rf $out && mkdir -p $out/obj && cp $ROOT/syllable_detector_swift_amd/lib/obj/*.o $out/obj/
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -D$v -I$ROOT/include -c $CS/kernels_fused_s.hip -o $out/obj/kernels_fused_s.hip.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libsyldet.so $out/obj/*.o
  echo built lib_$name
done
