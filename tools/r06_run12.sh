cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "grad or bwd or train or adam" 2>&1 | tail -3
for L in deepcgp_amd/ab/libdcgp_base.so deepcgp_amd/libdcgp.so deepcgp_amd/ab/libdcgp_base.so deepcgp_amd/libdcgp.so; do echo "== $L"; DCGP_LIB=$L timeout 300 python tools/grad_time.py cfg2_mnist_CH_M256 40 2>&1 | tail -2; DCGP_DEDUP=1 DCGP_LIB=$L timeout 300 python tools/grad_time.py cfg2_mnist_CH_M256 40 2>&1 | tail -2; done
