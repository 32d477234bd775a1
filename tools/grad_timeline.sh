#!/bin/bash
# usage (GPU box, repo root): tools/grad_timeline.sh <tag> [config] [steps]  -- every launch of ONE training step (value + gradient) with stream and start time
TAG=$1; shift
tools/prof_grad.sh $TAG "$@" > gpurun_out/${TAG}_grad_step_summary.txt 2>&1
DB=$(find gpurun_out/prof_$TAG -name '*.db' | head -1)
python - $DB > gpurun_out/${TAG}_grad_step_timeline.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = c.execute("select %s, start, end, %s from kernels order by start" % (name_col, qcol)).fetchall()
marks = [i for i, r in enumerate(rows) if "prepare_all" in r[0]]
lo, hi = marks[-3], marks[-2]
t0 = rows[lo][1]
for r in rows[lo:hi]:
    nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:56]
    print("%9.1f us  +%8.1f us  q=%s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], nm))
print("step span: %.1f us, %d launches" % ((rows[hi][1] - t0) / 1e3, hi - lo))
PY
rm -rf gpurun_out/prof_$TAG
