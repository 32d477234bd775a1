cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "fused" 2>&1 | tail -3
export FUSED_AB_SETS="fused_persist=0 fused_shape=2,fused_persist=2,fused_stagger=0 fused_shape=2,fused_persist=1,fused_stagger=0 fused_shape=2,fused_persist=1,fused_stagger=20 fused_shape=2,fused_persist=1,fused_stagger=40 fused_shape=2,fused_persist=1,fused_stagger=90 fused_shape=0,fused_persist=1 fused_persist=0"
timeout 600 python tools/fused_ab.py 2>&1 | tail -12
echo "=== shape 2 persistent dynamic stagger 0"; DCGP_FUSED_SHAPE=2 DCGP_FUSED_PERSIST=1 DCGP_FUSED_STAGGER=0 timeout 200 python tools/fused_trace.py --summary 2>&1 | head -80
