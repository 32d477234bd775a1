export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1; grep -n "passed\|failed" gpurun_out/t.log
bash tools/prof_grad.sh r02_grad > gpurun_out/r02_grad_step_summary_raw.txt 2>&1
DB=$(find gpurun_out/prof_r02_grad -name "*.db" | head -1); python tools/rocpd_timeline.py $DB -3 > gpurun_out/r02_grad_step_timeline.txt 2>/dev/null
DCGP_DEDUP=1 bash tools/prof_grad.sh r02_grad_dedup > gpurun_out/r02_grad_step_dedup_summary_raw.txt 2>&1
DB=$(find gpurun_out/prof_r02_grad_dedup -name "*.db" | head -1); python tools/rocpd_timeline.py $DB -3 > gpurun_out/r02_grad_step_dedup_timeline.txt 2>/dev/null
python tools/grad_time.py cfg2_mnist_CH_M256 50 2>&1 | tail -2
DCGP_DEDUP=1 python tools/grad_time.py cfg2_mnist_CH_M256 50 2>&1 | tail -2
python bench.py --steps 200 --warmup 20 --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'grad' in k or 'train' in k})"
