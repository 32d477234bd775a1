# round-5 head-sweep A/B on the GPU box: parity tests of the sweeps, placement traces, the sweep table, a short bench of the named configs
export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-x}
python -m pytest tests -m gpu -x -q -k "head or kuf or sweep or golden_ops or baseline_configs" > gpurun_out/head_${tag}_tests.txt 2>&1; tail -3 gpurun_out/head_${tag}_tests.txt
for c in cfg2_mnist_CH_M256 cfg3_mnist_3layer_M256 cfg4_cifar_3layer_M384; do python tools/sweep_trace.py $c head_sweep; done > gpurun_out/head_${tag}_place.txt 2>&1
for c in cfg3_mnist_3layer_M256 cfg4_cifar_3layer_M384; do python tools/sweep_trace.py $c kuf_long; done >> gpurun_out/head_${tag}_place.txt 2>&1
grep -E "head_sweep:|kuf_long:|late waves|first unit" gpurun_out/head_${tag}_place.txt
for c in cfg2_mnist_CH_M256 cfg3_mnist_3layer_M256 cfg4_cifar_3layer_M384; do
  python bench.py --config $c --steps 100 --no-cpu-baseline --no-grad-leg --no-extra-legs --no-all-configs > gpurun_out/head_${tag}_bench_$c.json 2> gpurun_out/head_${tag}_bench_$c.err
  python - <<PY
import json
d=json.load(open("gpurun_out/head_${tag}_bench_$c.json"))
print("$c", d["value"], d["ms_per_step"], {k:v for k,v in d.get("kernel_times_us",{}).items()} if "kernel_times_us" in d else "")
PY
done
