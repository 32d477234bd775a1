#!/bin/bash
# usage (GPU box, repo root): tools/pmc_grad.sh <tag> "<counters>" [config]   -- counters of one value+gradient step, own pass
TAG=$1; CNT=$2; shift; shift
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out
export DCGP_NO_SIDE_STREAM=1   # counter mode serialises dispatches; cross-stream waits deadlock it
cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $CNT -d $R/gpurun_out/pmc_$TAG -o pmc --output-format csv -- python $R/tools/grad_time.py ${1:-cfg2_mnist_CH_M256} 1 > $R/gpurun_out/pmc_${TAG}.log 2>&1
cd $R
F=$(find gpurun_out/pmc_$TAG -name '*counter_collection.csv' | head -1)
python - "$F" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('::')[-1].split('(')[0][:44] + ' g=' + r.get('Grid_Size', '?')
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (k, r['Dispatch_Id'])
    if key not in seen:
        seen.add(key); cnt[k] += 1
first = lambda k: max(agg[k].values())
for k in sorted(agg, key=lambda k: -first(k))[:8]:
    print(k, 'dispatches', cnt[k])
    for c, v in sorted(agg[k].items()):
        print('    %-32s %.4g per dispatch' % (c, v / cnt[k]))
PY
