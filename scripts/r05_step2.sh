#!/bin/bash
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_primitives.py -x -q -k "gemm" > $O/t_gemm.txt 2>&1; echo "gemm rc $?" >> $O/t_gemm.txt
tail -3 $O/t_gemm.txt
for D in 1 0 2; do
  DM_GEMM_DMA=$D timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-h2d-leg --pmc-json /nonexistent --shape-table $O/shapes_dma$D.txt > $O/bench_dma$D.json 2> $O/bench_dma$D.err
  python - <<PY
import json
d=json.load(open('$O/bench_dma$D.json')); print('DMA=$D ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline'].get('all_gemm'))
PY
done
