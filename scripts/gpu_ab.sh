#!/bin/bash
# A/B of one environment switch on the default bench (diagnostic): bash scripts/gpu_ab.sh DM_MLP_NO_SPARSE [pytest -k expression]
O=gpurun_out/ab; mkdir -p $O
export PYTHONPATH=$PWD
if [ -n "$2" ]; then timeout 900 python -m pytest tests/test_gpu_training_step.py -m gpu -q -x -k "$2" > $O/pt.log 2>&1; tail -4 $O/pt.log; fi
timeout 200 python bench.py --no-h2d-leg --no-cpu-baseline --steps 40 --shape-table $O/shapes_on.txt > $O/bench_on.json 2> $O/bench_on.err
env $1=1 timeout 200 python bench.py --no-h2d-leg --no-cpu-baseline --steps 40 --shape-table $O/shapes_off.txt > $O/bench_off.json 2> $O/bench_off.err
python - <<'PY'
import json
for f in ('bench_on','bench_off'):
    d=json.loads(open(f'gpurun_out/ab/{f}.json').read().strip().splitlines()[-1])
    print(f, round(d['value'],3), round(d['ms_per_step'],3), d['loss_model_last'])
PY
