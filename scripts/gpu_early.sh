#!/bin/bash
# early head window (DM_HEADS_EARLY): tests, then the bench with / without, fp32 / bf16 / shard
O=gpurun_out/early; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests -m gpu -q -x -k "early_head or deferred or two_steps or goldens or literal or gae or dream or graphed or mlp or dmc or iwae" > $O/pt.log 2>&1; tail -3 $O/pt.log
p() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), d.get('loss_model_last'))"; }
for dt in f32 bf16; do
  for v in 1 0; do
    DM_HEADS_EARLY=$v timeout 200 python bench.py --dtype $dt --no-h2d-leg --no-cpu-baseline --steps 40 --prof-steps 0 2>/dev/null | p ${dt}_early$v
  done
done
for v in 1 0; do DM_HEADS_EARLY=$v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 --emulate-world 8 2>/dev/null | p shard_early$v; done
