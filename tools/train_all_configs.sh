#!/bin/bash
# usage (GPU box, repo root): tools/train_all_configs.sh > out.txt  -- the training step of every BASELINE configuration with a conv layer or M >= 256,
# with the exact layer-0 de-duplication (DCGP_DEDUP=1: what models.train() runs) and on the tiled batch: value + gradient, and value + gradient + Adam in one call
for c in cfg2_mnist_CH_M256 cfg2_mnist_H_M256 cfg3_mnist_3layer_M256 cfg4_cifar_3layer_M384 cfg5_mnist_H_M1024 cfg5_mnist_CH_M1024; do
  for d in 1 0; do
    echo "== $c  DCGP_DEDUP=$d"
    DCGP_DEDUP=$d python tools/grad_time.py $c 20 2>&1 | grep "ms/step"
  done
done
