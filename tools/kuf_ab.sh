# usage (GPU box): bash tools/kuf_ab.sh   -- the storing sweep under its A/B options, every configuration with a conv layer
for cfg in cfg2_mnist_CH_M256 cfg3_mnist_3layer_M256 cfg4_cifar_3layer_M384 cfg5_mnist_CH_M1024; do
for env in "" "DCGP_KUF_SPLIT=0" "DCGP_KUF_SPLIT=3" "DCGP_KUF_SPLIT=4" "DCGP_KUF_WPG=4" "DCGP_KUF_WPG=1"; do
  echo "== $cfg [$env]"; env $env python tools/sweep_times.py $cfg 2>&1 | grep "kuf "
done; done
