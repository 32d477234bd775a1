#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_grad.sh <tag> [config] [steps]
# rocprofv3 kernel trace of tools/grad_time.py -> per-kernel CSV summary gpurun_out/prof_<tag>_kernel_stats.csv
TAG=$1; shift
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o grad -- python $R/tools/grad_time.py "$@" > $R/gpurun_out/prof_${TAG}.log 2>&1
cd $R
DB=$(find gpurun_out/prof_$TAG -name '*.db' | head -1)
python tools/rocpd_summary.py $DB gpurun_out/prof_${TAG}_kernel_stats.csv
grep 'value+grad' gpurun_out/prof_${TAG}.log
python - "$TAG" <<'PY'
import csv, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open('gpurun_out/prof_%s_kernel_stats.csv' % tag)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:28]:
    print('%-40s grid %10s calls %5s avg_us %9.1f share %5.1f%%' % (r['Name'].split('::')[-1].split('(')[0][:40], r['GridThreads'], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
