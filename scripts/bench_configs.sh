#!/bin/bash
# whole-step numbers for a set of configurations: bash scripts/bench_configs.sh [extra bench.py flags]
for cfg in "--dtype f32" "--dtype bf16" "--dtype f32 --emulate-world 8" "--dtype bf16 --emulate-world 8" "--dtype f32 --emulate-world 2" "--dtype f32 --emulate-world 4"; do
  echo "== $cfg $*"
  timeout 300 python bench.py --reps 1 --steps 30 --warmup 8 --no-cpu-baseline --no-h2d-leg --prof-steps 0 $cfg "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   value', round(d['value'],3), d['unit'], 'ms_per_step', round(d['ms_per_step'],3), 'loss', d['loss_model_last'])
"
done
