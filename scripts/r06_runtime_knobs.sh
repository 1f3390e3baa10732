#!/bin/bash
# HIP runtime switches (names read from libamdhip64.so's flag table) against the step: queue management and kernel-argument placement
set -u
OUT=$PWD/gpurun_out/r06; mkdir -p $OUT
B="--steps 30 --warmup 8 --reps 3 --no-cpu-baseline --no-h2d-leg --prof-steps 0"
line() { python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', round(d['ms_per_step'],3), [round(x,3) for x in d['ms_per_step_regions']], 'host free', d.get('host_enqueue_ms_per_step_unthrottled'))
"; }
{
for cfg in "--dtype f32" "--dtype f32 --pipeline --emulate-world 8" "--dtype bf16"; do
  for V in "DM_X=0" "DEBUG_HIP_DYNAMIC_QUEUES=0" "DEBUG_HIP_DYNAMIC_QUEUES=0 GPU_MAX_HW_QUEUES=8" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "ROC_USE_FGS_KERNARG=0" "DEBUG_HIP_FORCE_ASYNC_QUEUE=1" "GPU_STREAMOPS_CP_WAIT=0"; do
    echo "== $cfg | $V"; env $V python bench.py $B $cfg 2>/dev/null | line
  done
done
for cfg in "--dtype f32 --emulate-world 2"; do
  for V in "DM_X=0" "DEBUG_HIP_DYNAMIC_QUEUES=0" "DEBUG_HIP_DYNAMIC_QUEUES=0 DM_BENCH_DP_IDLE=1" "DEBUG_HIP_DYNAMIC_QUEUES=0 DM_DP_EARLY=1"; do
    echo "== $cfg --force-dp | $V"; env $V python bench.py $B $cfg --force-dp 2>/dev/null | line
  done
  echo "== $cfg (no DP) | DEBUG_HIP_DYNAMIC_QUEUES=0"; env DEBUG_HIP_DYNAMIC_QUEUES=0 python bench.py $B $cfg 2>/dev/null | line
done
} > $OUT/r06_runtime_knobs.txt 2>&1
cat $OUT/r06_runtime_knobs.txt
