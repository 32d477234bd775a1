cd $GRAFT_REPO_ROOT
export FUSED_AB_SETS="fused_persist=0 fused_persist=0 fused_shape=0,fused_persist=1 fused_shape=2,fused_persist=1,fused_stagger=0 fused_shape=2,fused_persist=0"
for v in "" d3 d1 ""; do echo "== ${v:-default}"; L=deepcgp_amd/libdcgp.so; [ -n "$v" ] && L=deepcgp_amd/ab/libdcgp_$v.so; DCGP_LIB=$L timeout 300 python tools/fused_ab.py 2>&1 | tail -5 | cut -c1-110; done
