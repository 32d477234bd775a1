# usage (GPU box): bash tools/head_ab.sh   -- the reducing sweep (ConvKernel.Kzx + Kdiag) under the tail-balance option, every configuration
for cfg in cfg2_mnist_H_M256 cfg1_mnist_H_M32 cfg5_mnist_H_M1024 cfg2_mnist_CH_M256 cfg3_mnist_3layer_M256 cfg4_cifar_3layer_M384; do
for env in "" "DCGP_HEAD_TAIL=0" "DCGP_HEAD_TAIL=2" "DCGP_HEAD_TAIL=8"; do
  echo "== $cfg [$env]"; env $env python tools/sweep_times.py $cfg 2>&1 | grep "head_sweep"
done; done
