for u in 1 2 4; do DCGP_KUF_UPW=$u DCGP_NO_FUSED_LAYER=1 python bench.py --steps 30 --no-cpu-baseline --no-grad-leg --no-extra-legs 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('upw $u', round(d['value'],1), d['kernel_times_us'].get('kuf'))"; done
for u in 1 2 4; do DCGP_KUF_UPW=$u python bench.py --config cfg5_mnist_CH_M1024 --steps 10 --no-cpu-baseline --no-grad-leg --no-extra-legs 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('cfg5 upw $u', round(d['value'],2), d['kernel_times_us'].get('kuf'))"; done
