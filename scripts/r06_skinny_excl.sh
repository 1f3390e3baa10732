#!/bin/bash
# the <= 64-row chain kernels with LDS claimed beyond their need (no 48-KB tile workgroup fits beside a strip): two library builds
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
  for V in "DM_X=0" "DM_LIB_PATH=$PWD/pydreamer_amd/libdreamer_hip_skpad10240.so" "DM_LIB_PATH=$PWD/pydreamer_amd/libdreamer_hip_skpad20000.so"; do
    echo "== $cfg | ${V##*/}"; env $V python bench.py $B $cfg 2>/dev/null | line
  done
done
} > $OUT/r06_skinny_excl.txt 2>&1
cat $OUT/r06_skinny_excl.txt
