#!/bin/bash
# Diagnostic: variant builds of ONE kernel file with a define (a piece knocked out: results wrong by construction, or an
# experiment), each as its own library directory syllable_detector_swift_amd/lib_<name>/ for tools/ab_kernel.py / c3_timing.py:
#     tools/knockouts.sh kernels_bdft.hip SYLDET_B_NOWINDOW SYLDET_B_NOMFMA
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/syllable_detector_swift_amd/csrc
FILE=$1; shift
make -C $CS -j6 >/dev/null
for v in "$@"; do
  name=$(echo $v | tr 'A-Z' 'a-z' | sed 's/syldet_[a-z]_//; s/=/_/')
  out=$ROOT/syllable_detector_swift_amd/lib_$name
  rm -rf $out && mkdir -p $out/obj && cp $ROOT/syllable_detector_swift_amd/lib/obj/*.o $out/obj/
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -D$v -I$ROOT/include -c $CS/$FILE -o $out/obj/$FILE.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libsyldet.so $out/obj/*.o
  echo built lib_$name
done
