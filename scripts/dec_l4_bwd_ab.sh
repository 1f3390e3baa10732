#!/bin/bash
# the image layer's backward: direct MFMA kernels (default) against the gather-form products (DM_DEC_L4_BWD_GEMM=1)
O=gpurun_out/l4bwd; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_training_step.py -q -k "conv_decoder or conv_stack or bf16_storage or deferred" > $O/t_conv.txt 2>&1; echo "conv rc $?" | tee -a $O/t_conv.txt
tail -3 $O/t_conv.txt
for rep in 1 2; do
  for v in 0 1; do
    for cfg in "--dtype f32" "--dtype bf16"; do
      env $( [ $v = 1 ] && echo DM_DEC_L4_BWD_GEMM=1 || echo DM_X=0 ) python bench.py --reps 1 $cfg --steps 30 --warmup 8 --no-cpu-baseline --no-h2d-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('gemm-path $v  $cfg ', round(d['ms_per_step'],3), 'ms')"
    done
  done
done 2>&1 | tee $O/ab.txt
