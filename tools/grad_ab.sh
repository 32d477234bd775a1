#!/bin/bash
# usage (GPU box, repo root): tools/grad_ab.sh "<ENV=1 ...>" ...  -- the training step (cfg2 CH, de-duplicated and not) under each environment setting
for e in "" "$@"; do
  for d in 1 0; do
    echo "== [$e] dedup=$d"
    env $e DCGP_DEDUP=$d python tools/grad_time.py cfg2_mnist_CH_M256 30
  done
done
