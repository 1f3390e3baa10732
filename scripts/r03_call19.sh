#!/bin/bash
O=gpurun_out/r03s; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python -m pytest tests/test_gpu_training_step.py -m gpu -q -x -k "persistent or rssm_sequence" > $O/pt.log 2>&1; tail -15 $O/pt.log
timeout 200 python bench.py --no-h2d-leg --no-cpu-baseline --steps 30 > $O/bench_on.json 2> $O/bench_on.err
DM_RSSM_PERSIST=0 timeout 200 python bench.py --no-h2d-leg --no-cpu-baseline --steps 30 > $O/bench_off.json 2> $O/bench_off.err
python - <<'PY'
import json
for f in ('bench_on','bench_off'):
    try:
        d=json.loads(open(f'gpurun_out/r03s/{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['loss_model_last'])
    except Exception as e:
        print(f, 'failed', e)
PY
