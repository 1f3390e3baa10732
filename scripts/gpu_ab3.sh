#!/bin/bash
O=gpurun_out/ab3; mkdir -p $O
export PYTHONPATH=$PWD
for v in off on; do
  if [ $v = on ]; then export DM_GEMM_ACC2=1; fi
  timeout 200 python scripts/gemm_modes.py --only 8,9,10 > $O/modes_$v.txt 2>&1
  timeout 200 python bench.py --no-h2d-leg --no-cpu-baseline --steps 40 > $O/bench_$v.json 2> $O/bench_$v.err
done
python - <<'PY'
import json
for v in ('off','on'):
    for ln in open(f'gpurun_out/ab3/modes_{v}.txt'):
        if ln.startswith('{'):
            d=json.loads(ln); print(v, d['what'], round(d['us'],1), round(d['tflops'],1), d['err']['rel_l2'])
    d=json.loads(open(f'gpurun_out/ab3/bench_{v}.json').read().strip().splitlines()[-1])
    print(v, 'bench', round(d['ms_per_step'],3), round(d['roofline']['frac'],3), d['loss_model_last'])
PY
