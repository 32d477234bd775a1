#!/bin/bash
# usage (GPU box): tools/ab_cfgs.sh "<tags>" <config>...   -- one bench line per (library, config)
TAGS=$1; shift
for C in "$@"; do for T in $TAGS; do
  if [ "$T" = main ]; then unset DCGP_LIB; else export DCGP_LIB=$PWD/deepcgp_amd/ab/libdcgp_$T.so; fi
  python bench.py --config $C --steps 40 --no-cpu-baseline --no-grad-leg --no-extra-legs 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$T', d['config']['workload'], round(d['value'],1), round(d['ms_per_step'],4), round(d['steps_per_s_two_in_flight'],1), d['kernel_times_us'])"
done; done
