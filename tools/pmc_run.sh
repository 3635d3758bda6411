#!/bin/bash
# usage: tools/pmc_run.sh <outdir-name> "<counters>" [bench args...]   (run on the GPU box via gpurun)
# Collects PMC counters for bench.py's kernels in their own pass (no trace domains besides --kernel-trace).
set -e
name=$1; shift
ctrs=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/bench.log 2>&1 || true
cd $GRAFT_REPO_ROOT
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/**/*counter_collection.csv', recursive=True)
if not f:
    print('no counter file'); print(open(out + '/bench.log').read()[-2000:]); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row['Kernel_Name'][:60]
    agg[k][row['Counter_Name']] += float(row['Counter_Value'])
    cnt[(k, row['Counter_Name'])] += 1
for k, d in agg.items():
    if 'sd::' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        n = cnt[(k, c)]
        print('   %-28s per-dispatch %.6g  (dispatches %d)' % (c, v / n, n))
PY
