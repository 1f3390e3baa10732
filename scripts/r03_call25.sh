#!/bin/bash
O=gpurun_out/r03y; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_training_step.py -m gpu -q -x -k "mlp_head or dream or goldens or atari_literal or full_size or two_steps" > $O/pt.log 2>&1; tail -5 $O/pt.log
for v in on off; do
  if [ $v = off ]; then export DM_MLP_NO_SPARSE=1; fi
  timeout 200 python bench.py --no-h2d-leg --no-cpu-baseline --steps 40 > $O/bench_$v.json 2> $O/bench_$v.err
done
python - <<'PY'
import json
for f in ('bench_on','bench_off'):
    d=json.loads(open(f'gpurun_out/r03y/{f}.json').read().strip().splitlines()[-1])
    k=[x for x in d['roofline']['kinds'] if 'mlp_chain' in x['kernel']][0]
    print(f, d['value'], d['ms_per_step'], 'chain', k['launches_per_step'], round(k['avg_launch_us'],1), round(k['ms_per_step'],3))
PY
