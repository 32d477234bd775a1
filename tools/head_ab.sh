# usage (GPU box): bash tools/head_ab.sh   -- the patch sweeps under the occupancy-shaping option, every configuration
for cfg in cfg2_mnist_H_M256 cfg1_mnist_H_M32 cfg5_mnist_H_M1024 cfg2_mnist_CH_M256 cfg3_mnist_3layer_M256 cfg4_cifar_3layer_M384 cfg5_mnist_CH_M1024; do
for env in "" "DCGP_SWEEP_OCC=0" "DCGP_SWEEP_OCC=3" "DCGP_SWEEP_OCC=2"; do
  echo "== $cfg [$env]"; env $env python tools/sweep_times.py $cfg 2>&1 | grep "head_sweep\|kuf"
done; done
