export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "potrf or trtri or chain or kuu or conditional or golden or smoke" 2>&1 | tail -5
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCHOL_TRACE -I deepcgp_amd/csrc tools/chol_trace.hip -o /tmp/chol_trace -L deepcgp_amd -ldcgp -Wl,-rpath,$PWD/deepcgp_amd 2>/dev/null && /tmp/chol_trace 256 4 > gpurun_out/r05_chol_trace_halves.txt 2>&1
head -9 gpurun_out/r05_chol_trace_halves.txt
timeout 600 python bench.py --steps 100 --no-cpu-baseline --no-grad-leg --no-extra-legs > gpurun_out/r05_bench_halves.json 2> gpurun_out/r05_bench_halves.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_halves.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('head_only_steps_per_s'), d['kernel_times_us'])
PY
