#!/bin/bash
# quick A/B line: bash scripts/r06_quick.sh [label]  ->  step times of the three configurations
line() { python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), 'max', round(d['ms_per_step_max'],3), 'loss', d['loss_model_last'])
"; }
for cfg in ${CFGS:-"--dtype f32|--dtype bf16|--dtype f32 --pipeline --emulate-world 8"}; do :; done
IFS='|' read -ra CF <<< "${CFGS:---dtype f32|--dtype bf16|--dtype f32 --pipeline --emulate-world 8}"
for cfg in "${CF[@]}"; do
  echo "== $1 $cfg"
  timeout 300 python bench.py --steps 30 --warmup 8 --reps 3 --no-cpu-baseline --no-h2d-leg --prof-steps 0 $cfg 2>/dev/null | line
done
