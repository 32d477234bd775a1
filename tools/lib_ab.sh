#!/bin/bash
# usage (GPU box): tools/lib_ab.sh <config> <tag> [<tag> ...]  -- the layer kernel's launch and the synchronous step under A/B builds of the library
# (deepcgp_amd/ab/libdcgp_<tag>.so from tools/ab_build.sh; "main" = the shipped library), two passes over the list so that drift shows
CFG=$1; shift
for pass in 1 2; do
  for t in "$@"; do
    if [ "$t" = main ]; then unset DCGP_LIB; else export DCGP_LIB=deepcgp_amd/ab/libdcgp_$t.so; fi
    echo -n "$t: "
    FUSED_AB_SETS="${FUSED_AB_SETS:-fused_pre=-1}" python tools/fused_ab.py $CFG | tail -1
  done
done
