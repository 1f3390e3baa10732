#!/bin/bash
# posterior-chain timing with / without one switch: bash scripts/gpu_tloop.sh DM_Z_EMBED_NO_WIDE
O=gpurun_out/tloop; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python -m pytest tests/test_gpu_training_step.py -m gpu -q -x -k "persistent or rssm_sequence or goldens or two_steps" > $O/pt.log 2>&1; tail -2 $O/pt.log
for b in 50 7; do
  timeout 200 python scripts/persist_prof.py $b 2>/dev/null | grep "persist=0" | sed "s/^/on  /"
  env $1=1 timeout 200 python scripts/persist_prof.py $b 2>/dev/null | grep "persist=0" | sed "s/^/off /"
done
timeout 200 python bench.py --no-h2d-leg --no-cpu-baseline --steps 40 > $O/bench_on.json 2>/dev/null
env $1=1 timeout 200 python bench.py --no-h2d-leg --no-cpu-baseline --steps 40 > $O/bench_off.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench_on','bench_off'):
    d=json.loads(open(f'gpurun_out/tloop/{f}.json').read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3), d['loss_model_last'])
PY
