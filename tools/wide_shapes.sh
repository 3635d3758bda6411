#!/bin/bash
# A/B of the wide engine's two MFMA shapes on one box (config5): the shipped 16x16x32 against SYLDET_WIDE_SHAPE32=1.
for rep in 1 2; do
for v in "" "SYLDET_WIDE_SHAPE32=1"; do
  env $v python bench.py --workload config5 --no-also --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); f=r['roofline']
print('${v:-16x16x32 (default)}', 'ms_per_step %.2f'%r['ms_per_step'], f.get('kernel'), 'ms', '%.2f' % f['kernel_ms'][f['kernel']], 'frac %.3f'%f['frac'], 'achieved %.0f %s'%(f['achieved'], f['unit']))
"
done; done
