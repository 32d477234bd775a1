#!/bin/bash
# usage (GPU box, repo root): tools/prof_pipe.sh <tag> [config] [steps] [depth]  -- kernel trace of steps kept in flight + one step's timeline
TAG=$1; shift
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out
cd /tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/pipe_$TAG -o pipe -- python $R/tools/pipe_steps.py "$@" > $R/gpurun_out/pipe_${TAG}.log 2>&1
cd $R
tail -1 gpurun_out/pipe_${TAG}.log
DB=$(find gpurun_out/pipe_$TAG -name '*.db' | head -1)
python tools/rocpd_timeline.py $DB -4 > gpurun_out/pipe_${TAG}_timeline.txt
cat gpurun_out/pipe_${TAG}_timeline.txt
