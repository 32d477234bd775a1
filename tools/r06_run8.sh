cd $GRAFT_REPO_ROOT
export FUSED_AB_SETS="fused_persist=0 fused_persist=0"
for v in z0p0 z0p1 z0p2 z1p0 z1p1 z1p2 z0p0; do echo "== $v"; DCGP_LIB=deepcgp_amd/ab/libdcgp_$v.so timeout 300 python tools/fused_ab.py 2>&1 | tail -2; done
