#!/bin/bash
# Diagnostic: the fold kernel's padded sample ring (hops that are multiples of 64) against builds without the padding
# (lib_nopad: fused_r_kernel takes 256-sample windows there, the plain ring 128-sample ones) and without the slot permutation
# (lib_noperm), interleaved on ONE box.  Libraries: syllable_detector_swift_amd/lib_*/ (tools/knockouts.sh-style builds).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
for lib in lib lib_noperm lib_nopad; do
  for row in "N=256 hop 64" "N=128 hop 64" "N=256 hop 68"; do
    SYLDET_LIB=$ROOT/syllable_detector_swift_amd/$lib/libsyldet.so python3 $ROOT/tools/variant_timing.py --short --only "$row" 2>/dev/null | sed "s/^/$lib  /"
  done
  SYLDET_LIB=$ROOT/syllable_detector_swift_amd/$lib/libsyldet.so python3 $ROOT/tools/variant_timing.py --only "hop 128" 2>/dev/null | grep -v "H=8" | sed "s/^/$lib  /"
done; done
