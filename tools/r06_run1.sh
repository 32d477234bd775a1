set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "fused" 2>&1 | tail -5
echo "=== baseline lib"
DCGP_LIB=deepcgp_amd/ab/libdcgp_base.so FUSED_AB_SETS="fused_shape=-1 fused_shape=2 fused_shape=-1" timeout 300 python tools/fused_ab.py 2>&1 | tail -5
echo "=== new lib"
timeout 600 python tools/fused_ab.py 2>&1 | tail -12
