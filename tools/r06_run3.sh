cd $GRAFT_REPO_ROOT
echo "=== shape 2 non-persistent"; DCGP_FUSED_SHAPE=2 timeout 200 python tools/fused_trace.py 2>&1 | head -60
echo "=== shape 2 persistent stagger 0"; DCGP_FUSED_SHAPE=2 DCGP_FUSED_PERSIST=1 DCGP_FUSED_STAGGER=0 timeout 200 python tools/fused_trace.py 2>&1 | head -80
echo "=== shape 2 persistent stagger 40"; DCGP_FUSED_SHAPE=2 DCGP_FUSED_PERSIST=1 DCGP_FUSED_STAGGER=40 timeout 200 python tools/fused_trace.py 2>&1 | head -80
