#!/bin/bash
# What do the data-parallel collectives (and the stream they run on) cost a step?  A ONE-rank RCCL group on a 1-GPU box
# (bench.py --force-dp): no DP | the default (LATE: every group all-reduced inside grad_clip) | DM_DP_EARLY=1 (behind each pre-launched
# backward, overlapped, torch's communication stream) | the library's own dm_allreduce_grads (DM_DP_NATIVE=1), late and early
line() { python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), 'max', round(d['ms_per_step_max'],3), 'loss', d['loss_model_last'])
"; }
B="--steps 30 --warmup 8 --reps 3 --no-cpu-baseline --no-h2d-leg --prof-steps 0"
for cfg in "--dtype f32" "--dtype f32 --pipeline --emulate-world 8" "--dtype bf16 --pipeline --emulate-world 8"; do
  echo "== $cfg | no DP";                         timeout 300 python bench.py $B $cfg 2>/dev/null | line
  echo "== $cfg | --force-dp (late, default)";    timeout 300 python bench.py $B $cfg --force-dp 2>/dev/null | line
  echo "== $cfg | --force-dp DM_DP_EARLY=1";      DM_DP_EARLY=1 timeout 300 python bench.py $B $cfg --force-dp 2>/dev/null | line
  echo "== $cfg | --force-dp native late";        DM_DP_NATIVE=1 timeout 300 python bench.py $B $cfg --force-dp 2>/dev/null | line
done
