#!/bin/bash
# A/B of the persistent BPTT kernel inside the whole step (fp32, bf16, 7-column shard), then the chains alone
mkdir -p gpurun_out
for v in 1 0; do
  for cfg in "--dtype f32" "--dtype bf16" "--dtype f32 --emulate-world 8" "--dtype bf16 --emulate-world 8"; do
    echo "== DM_RSSM_LDS_BWD=$v $cfg"
    DM_RSSM_LDS_BWD=$v timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-h2d-leg --prof-steps 0 $cfg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   value', round(d['value'],3), d['unit'], 'ms_per_step', round(d['ms_per_step'],3), 'loss', d['loss_model_last'])
"
  done
done
timeout 300 python scripts/persist_prof.py 50 7 2>&1 | grep -v Warning | grep -v "per step" | tail -12
