#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_bench.sh <tag> [bench args...]
# rocprofv3 kernel trace of bench.py -> gpurun_out/prof_<tag>/ + a per-kernel CSV summary (tools/rocpd_summary.py)
TAG=$1; shift
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --profile "$@" > $R/gpurun_out/prof_${TAG}_bench.log 2>&1
cd $R
DB=$(find gpurun_out/prof_$TAG -name '*.db' | head -1)
python tools/rocpd_summary.py $DB gpurun_out/prof_${TAG}_kernel_stats.csv
grep '^{' gpurun_out/prof_${TAG}_bench.log | tail -1 > gpurun_out/prof_${TAG}_bench.json
python - "$TAG" <<'PY'
import csv, json, sys
tag = sys.argv[1]
d = json.load(open('gpurun_out/prof_%s_bench.json' % tag))
steps = d['steps'] + d['warmup']   # bench.py --profile runs exactly warmup + steps steps
print('steps profiled', steps, 'ms/step', d['ms_per_step'])
tot = 0
for r in csv.DictReader(open('gpurun_out/prof_%s_kernel_stats.csv' % tag)):
    per = float(r['TotalDurationNs']) / steps / 1e3
    tot += per
    print('%-34s grid %9s calls/step %5.1f avg_us %8.1f per-step_us %8.1f' % (r['Name'].split('::')[-1].split('(')[0][:34], r['GridThreads'], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, per))
print('sum of kernel time per step (us): %.1f' % tot)
PY
