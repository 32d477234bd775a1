#!/bin/bash
# usage (GPU box, repo root): tools/pmc_bench.sh <tag> "<counters>" [bench args]   -- counters in their own pass (kernel-trace only)
TAG=$1; CNT=$2; shift; shift
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out
# one stream only: the profiler serialises dispatches in counter mode and cross-stream event waits deadlocked it
export DCGP_NO_SIDE_STREAM=1
cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $CNT -d $R/gpurun_out/pmc_$TAG -o pmc --output-format csv -- python $R/bench.py --profile "$@" > $R/gpurun_out/pmc_${TAG}.log 2>&1
cd $R
F=$(find gpurun_out/pmc_$TAG -name '*counter_collection.csv' | head -1)
python - "$F" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
rows = list(csv.DictReader(open(f)))
seen = set()
for r in rows:
    k = r['Kernel_Name'].split('::')[-1].split('(')[0][:40] + ' g=' + r.get('Grid_Size', '?')
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (k, r['Dispatch_Id'])
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in sorted(agg, key=lambda k: -agg[k].get('GRBM_GUI_ACTIVE', agg[k].get('SQ_WAVE_CYCLES', 0)))[:12]:
    print(k, 'dispatches', cnt[k])
    for c, v in sorted(agg[k].items()):
        print('    %-32s %.4g per dispatch' % (c, v / cnt[k]))
PY
