# usage: tools/step_timeline.sh <tag> <config> [env...]   -- step timeline of a bench config under rocprofv3
T=$1; C=$2; shift; shift
export TMPDIR=/tmp; mkdir -p gpurun_out
env "$@" tools/prof_bench.sh $T --config $C --steps 60 --warmup 20 --no-cpu-baseline --no-grad-leg --no-extra-legs > gpurun_out/${T}_summary.txt 2>&1
DB=$(find gpurun_out/prof_$T -name '*.db' | head -1)
python tools/rocpd_timeline.py $DB -4 > gpurun_out/${T}_timeline.txt
cat gpurun_out/${T}_timeline.txt
rm -rf gpurun_out/prof_$T
