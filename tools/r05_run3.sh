export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 600 python bench.py --steps 100 --no-cpu-baseline --no-grad-leg --no-extra-legs > gpurun_out/r05_bench_ride.json 2> gpurun_out/r05_bench_ride.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_ride.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('head_only_steps_per_s'), d['kernel_times_us'])
PY
