#!/bin/bash
# usage (GPU box): tools/ab_head.sh <config> <tag>...   -- head-only bench line per A/B library ("main" = the shipped one)
CFG=$1; shift
for T in "$@"; do
  if [ "$T" = main ]; then unset DCGP_LIB; else export DCGP_LIB=$PWD/deepcgp_amd/ab/libdcgp_$T.so; fi
  python bench.py --config $CFG --steps 50 --no-cpu-baseline --no-grad-leg --no-extra-legs 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$T', round(d['value'],1), round(d['ms_per_step'],4), round(d['steps_per_s_two_in_flight'],1), d['kernel_times_us'])"
done
