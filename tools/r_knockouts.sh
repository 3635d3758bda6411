#!/bin/bash
# Diagnostic (run on the GPU box via gpurun): kernel time of the benchmark batch with the shipped library ("lib") and with
# diagnostic builds of it that knock one piece of the register-resident-basis fused kernel out -- built beside it with
#   make -C syllable_detector_swift_amd/csrc OUT=$PWD/syllable_detector_swift_amd/lib_NOSTAGE \
#        CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DSYLDET_R_NOSTAGE"
# (NOSTAGE, NOEVAL, NOMAG, NOCARRY, NOMAX; NOLOAD = reloads from a cache-resident pass).  Their results are wrong by design.
# usage: tools/r_knockouts.sh lib lib_NOSTAGE ...
for v in "$@"; do
  SYLDET_LIB=$PWD/syllable_detector_swift_amd/$v/libsyldet.so python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['roofline']['kernel_ms'])"
done
