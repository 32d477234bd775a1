# quick A/B on the GPU box: parity tests of the head, short benches of the named configs
export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-x}; shift
python -m pytest tests -m gpu -x -q -k "head or svgp or golden or elbo or baseline_configs" > gpurun_out/small_${tag}_tests.txt 2>&1; tail -3 gpurun_out/small_${tag}_tests.txt
for c in "$@"; do
  python bench.py --config $c --steps 200 --no-cpu-baseline --no-grad-leg --no-extra-legs --no-all-configs > gpurun_out/small_${tag}_bench_$c.json 2> gpurun_out/small_${tag}_bench_$c.err
  python - <<PY
import json
d=json.load(open("gpurun_out/small_${tag}_bench_$c.json"))
print("$c", round(d["value"],1), round(d["ms_per_step"],4), d.get("kernel_times_us"))
PY
done
