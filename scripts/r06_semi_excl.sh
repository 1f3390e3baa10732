#!/bin/bash
# the exclusive whole-MLP build claiming less than the whole accumulator file (a127 / a191 instead of a255): room for ONE wave of another
# kernel beside the chain wave - less waiting for an empty SIMD, some co-residency.  Two library builds on one box.
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
for cfg in "--dtype f32" "--dtype bf16" "--dtype f32 --pipeline --emulate-world 8" "--dtype f32 --emulate-world 2"; do
  for V in "DM_X=0" "DM_LIB_PATH=$PWD/pydreamer_amd/libdreamer_hip_excl191.so" "DM_LIB_PATH=$PWD/pydreamer_amd/libdreamer_hip_excl127.so"; do
    echo "== $cfg | ${V##*/}"; env $V python bench.py $B $cfg 2>/dev/null | line
  done
done
} > $OUT/r06_semi_excl.txt 2>&1
cat $OUT/r06_semi_excl.txt
