# timeline of one training step (value + gradient) under rocprofv3: tools/grad_step_timeline.sh <tag> [env...]
T=$1; shift
export TMPDIR=/tmp; mkdir -p gpurun_out
env "$@" tools/prof_grad.sh $T cfg2_mnist_CH_M256 20 > gpurun_out/${T}_summary.txt 2>&1
DB=$(find gpurun_out/prof_$T -name '*.db' | head -1)
python tools/rocpd_timeline.py $DB -3 > gpurun_out/${T}_timeline.txt
rm -rf gpurun_out/prof_$T
wc -l gpurun_out/${T}_timeline.txt
