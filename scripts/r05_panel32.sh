#!/bin/bash
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_training_step.py -x -q -k "mlp_head" > $O/t_mlp.txt 2>&1; echo "mlp rc $?" >> $O/t_mlp.txt
tail -4 $O/t_mlp.txt
for P in 1 0 1 0; do
  DM_PANEL32=$P timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-h2d-leg --prof-steps 1 --pmc-json /nonexistent > $O/bench_p32_$P.json 2> $O/bench_p32_$P.err
  python - <<PY
import json
d=json.load(open('$O/bench_p32_$P.json')); k={x['kernel']:(round(x['ms_per_step'],2),round(x['tflops'],1)) for x in d['roofline']['kinds'] if 'panel' in x['kernel']}
print('PANEL32=$P ms/step', d['ms_per_step'], d['roofline']['all_gemm']['ms_per_step'], k, 'loss', d['loss_model_last'])
PY
done
