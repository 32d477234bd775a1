cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "fused or cfg2 or conv_layer" 2>&1 | tail -3
export FUSED_AB_SETS="fused_persist=0 fused_shape=0,fused_persist=1 fused_shape=2,fused_persist=1,fused_stagger=0 fused_persist=0"
timeout 600 python tools/fused_ab.py 2>&1 | tail -12
timeout 600 python tools/fused_ab.py cfg3_mnist_3layer_M256 2>&1 | tail -12
echo "=== shape 0"; timeout 200 python tools/fused_trace.py 2>&1 | head -30 | cut -c1-330
