cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "fused" 2>&1 | tail -2
export FUSED_AB_SETS="fused_persist=0 fused_persist=0 fused_shape=0,fused_persist=1"
for v in "" p50_80_95 ""; do echo "== ${v:-new}"; L=deepcgp_amd/libdcgp.so; [ -n "$v" ] && L=deepcgp_amd/ab/libdcgp_$v.so; DCGP_LIB=$L timeout 300 python tools/fused_ab.py 2>&1 | tail -3; done
timeout 200 python tools/fused_trace.py --summary 2>&1 | head -12
