export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/head_abl.txt; : > $out
run() { echo "=== $1" >> $out; python tools/sweep_trace.py cfg2_mnist_CH_M256 head_sweep 2>&1 | grep -E "span|late waves|first unit" >> $out; }
run base
cd deepcgp_amd/csrc
for f in HU_ABL_NOA HU_ABL_NOB "HU_ABL_NOA -DHU_ABL_NOB"; do
  rm -f head_units.o; make CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDCGP_EXPERIMENTS -D$f" > /dev/null 2>&1
  cd ../..; run "$f"; cd deepcgp_amd/csrc
done
cd ../..; cat $out
