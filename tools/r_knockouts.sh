#!/bin/bash
# Diagnostic: kernel time of the register-resident-basis fused kernel, the shipped build ("lib") and builds with parts
# knocked out (-DSYLDET_R_NO{LOAD,STAGE,EVAL,MAG,MAX} into syllable_detector_swift_amd/lib_<variant>/; wrong results by design).
for v in "$@"; do
  SYLDET_LIB=$PWD/syllable_detector_swift_amd/$v/libsyldet.so python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['roofline']['kernel_ms'])"
done
