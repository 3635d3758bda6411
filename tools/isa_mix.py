#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing: tools/isa_mix.py file.s substring"""
import sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l][0]
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start + 1:end]
c = Counter(l.split()[0] for l in body if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':'))
watch = ['v_mfma_f32_32x32x16_f16', 'global_load_dwordx4', 'global_load_dwordx2', 'global_load_dword', 'ds_read_b128', 'ds_read_b64',
         'ds_read2_b64', 'ds_read2st64_b64', 'ds_read_b32', 'ds_read2_b32', 'ds_write_b64', 'ds_write2_b64', 'ds_write_b32',
         'scratch_load_dword', 'scratch_store_dword', 's_waitcnt', 'v_cvt_pkrtz_f16_f32', 's_barrier', 'v_writelane_b32',
         'v_readlane_b32', 'v_accvgpr_write_b32', 'v_accvgpr_read_b32', 's_load_dword', 's_load_dwordx2', 's_load_dwordx4', 's_cbranch_scc1', 's_cbranch_vccz', 's_cbranch_execz']
for k in watch:
    if c.get(k):
        print('%-28s %d' % (k, c[k]))
print('total', sum(c.values()))
print(c.most_common(14))
