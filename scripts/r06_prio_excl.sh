#!/bin/bash
# stream priorities re-measured with the exclusive-SIMD whole-MLP build (round 6): the rollout's chain kernel needs an EMPTY SIMD, and the
# world-model stream is the high-priority one - does the rollout starve behind the decoder's backward products?
set -u
OUT=$PWD/gpurun_out/r06; mkdir -p $OUT
B="--steps 30 --warmup 8 --reps 3 --no-cpu-baseline --no-h2d-leg --prof-steps 0"
line() { python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', round(d['ms_per_step'],3), [round(x,3) for x in d['ms_per_step_regions']])
"; }
{
for cfg in "--dtype f32" "--dtype bf16" "--dtype f32 --pipeline --emulate-world 8"; do
  for V in "DM_X=0" "DM_WM_PRIO=0" "DM_WM_PRIO=0 DM_MAIN_PRIO=-1" "DM_MAIN_PRIO=-1" "DM_MAIN_PRIO=-1 DM_AC_PRIO=-1" "DM_WM_PRIO=0 DM_MAIN_PRIO=-1 DM_AC_PRIO=-1"; do
    echo "== $cfg | $V"; env $V python bench.py $B $cfg 2>/dev/null | line
  done
done
} > $OUT/r06_prio_excl.txt 2>&1
cat $OUT/r06_prio_excl.txt
