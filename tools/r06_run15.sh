cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "fused or persistent or cfg2 or cfg3 or baseline_configs" 2>&1 | tail -3
export FUSED_AB_SETS="fused_persist=0 fused_persist=-1 fused_persist=2 fused_persist=1 fused_persist=0 fused_persist=-1"
timeout 300 python tools/fused_ab.py 2>&1 | tail -6 | cut -c1-110
timeout 300 python tools/fused_ab.py cfg3_mnist_3layer_M256 2>&1 | tail -6 | cut -c1-110
