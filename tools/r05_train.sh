# training-step A/B on the GPU box: gradient parity tests, then the bench's training keys
export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-x}
python -m pytest tests -m gpu -x -q -k "grad or adam or autograd or train or natgrad or sgd" > gpurun_out/train_${tag}_tests.txt 2>&1; tail -3 gpurun_out/train_${tag}_tests.txt
python bench.py --no-cpu-baseline --no-all-configs --no-extra-legs > gpurun_out/train_${tag}_bench.json 2> gpurun_out/train_${tag}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/train_${tag}_bench.json"))
for k in ("value","value_and_grad_ms","train_step_ms_value_grad_adam","train_step_ms_with_exact_layer0_dedup","train_steps_per_s","train_steps_per_s_with_exact_layer0_dedup"): print(k, d.get(k))
print(d.get("roofline_train",{}).get("frac"))
PY
