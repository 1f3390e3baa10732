#!/bin/bash
# A/B of the world-model forward's tail on the world-model stream (default) against the caller's stream (DM_WM_TAIL=0)
O=gpurun_out/wm_tail; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_training_step.py -q -x -k "tail_on_side or bit_identical or literal_trainer or two_steps or older_step or packed or uint8" > $O/t.txt 2>&1; echo "rc $?" | tee -a $O/t.txt
tail -4 $O/t.txt
for rep in 1 2; do
  for v in 1 0; do
    for cfg in "--dtype f32" "--dtype bf16" "--dtype f32 --emulate-world 8" "--dtype f32 --emulate-world 2"; do
      DM_WM_TAIL=$v timeout 300 python bench.py --reps 1 $cfg --steps 30 --warmup 8 --no-cpu-baseline --no-h2d-leg --prof-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('tail $v  $cfg ', round(d['ms_per_step'],3), 'ms')"
    done
  done
done 2>&1 | tee $O/ab.txt
