#!/usr/bin/env python3
"""Instruction mix of the largest loop of one kernel in a hipcc -S listing:  tools/isa_loop.py file.s substring-of-the-symbol"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0]][0]
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
labels = {l.split(':')[0]: i for i, l in enumerate(body) if l.startswith('.LBB')}
best = None
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\S+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = i - labels[m.group(1)]
        if best is None or span > best[0]:
            best = (span, labels[m.group(1)], i)
_, a, b = best
loop = [l.split()[0] for l in body[a:b + 1] if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':')]
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_'): return 'valu'
    return 'other'
print("instructions in the largest loop:", len(loop), dict(Counter(cls(o) for o in loop)))
for op, n in Counter(loop).most_common(45):
    print("  %-28s %d" % (op, n))
