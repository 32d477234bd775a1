#!/bin/bash
# usage (GPU box, repo root): tools/bwd_frags_ab.sh  -- strip width of conv_bwd_fused (ctx option fused_bwd_frags): parity of every width, then the training step
for f in 4 2 1; do
  echo "== fused_bwd_frags=$f: gradient tests"
  DCGP_FUSED_BWD_FRAGS=$f python -m pytest tests/test_gpu_model.py -q -x -k "gradient" 2>&1 | tail -2
done
for f in 0 4 2 1; do
  for d in 1 0; do
    echo "== fused_bwd_frags=$f dedup=$d"
    DCGP_FUSED_BWD_FRAGS=$f DCGP_DEDUP=$d python tools/grad_time.py cfg2_mnist_CH_M256 30
  done
done
