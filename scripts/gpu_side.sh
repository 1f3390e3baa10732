#!/bin/bash
# deferred weight gradients (DM_WGRAD_SIDE): tests touching the decoder backward, then the bench with / without, fp32 and bf16
O=gpurun_out/side; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests -m gpu -q -x -k "deferred or rssm or goldens or two_steps or literal or gru or stack or iwae or graphed or chain" > $O/pt.log 2>&1; tail -3 $O/pt.log
for dt in f32 bf16; do
  for v in 1 0; do
    DM_WGRAD_SIDE=$v timeout 200 python bench.py --dtype $dt --no-h2d-leg --no-cpu-baseline --steps 40 --pmc-json /nonexistent > $O/bench_${dt}_$v.json 2> $O/bench_${dt}_$v.err
  done
done
DM_WGRAD_SIDE=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 --emulate-world 8 > $O/shard_1.json 2>/dev/null
DM_WGRAD_SIDE=0 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 --emulate-world 8 > $O/shard_0.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench_f32_1','bench_f32_0','bench_bf16_1','bench_bf16_0','shard_1','shard_0'):
    try:
        d=json.loads(open(f'gpurun_out/side/{f}.json').read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3), d.get('loss_model_last'))
    except Exception as e: print(f, 'failed', e)
PY
