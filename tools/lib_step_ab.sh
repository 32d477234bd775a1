#!/bin/bash
# usage (GPU box): tools/lib_step_ab.sh "<config ...>" <tag> [<tag> ...]  -- the synchronous forward ELBO step (tools/step_ab.py, default options) under A/B builds
# of the library (deepcgp_amd/ab/libdcgp_<tag>.so from tools/ab_build.sh; "main" = the shipped library), two passes over the list so that drift shows
CFGS=$1; shift
for pass in 1 2; do
  for t in "$@"; do
    if [ "$t" = main ]; then unset DCGP_LIB; else export DCGP_LIB=deepcgp_amd/ab/libdcgp_$t.so; fi
    STEP_AB_SETS="${STEP_AB_SETS:-fused_pre=-1}" python tools/step_ab.py $CFGS | awk -v t=$t 'NR%2==0 {print t": "$0}'
  done
done
