cd $GRAFT_REPO_ROOT
export FUSED_AB_SETS="fused_persist=0 fused_persist=0"
for v in "" p50_80_95 p60_85_95 p40_70_90 p50_75_95 p70_90_97 p30_60_85 ""; do echo "== ${v:-default 50 75 90}"; L=deepcgp_amd/libdcgp.so; [ -n "$v" ] && L=deepcgp_amd/ab/libdcgp_$v.so; DCGP_LIB=$L timeout 300 python tools/fused_ab.py 2>&1 | tail -2; done
