#!/bin/bash
mkdir -p gpurun_out/r05
O=gpurun_out/r05
for E in 0 1 2 0 1 2; do
  echo "== DM_GEMM_DMA_EARLY=$E"
  DM_GEMM_DMA_EARLY=$E python scripts/gemm_bench.py --only 0,1,2,9,10,22 --reps 20 2>&1 | grep -v Warn
done > $O/dma_early_gemm.txt
cat $O/dma_early_gemm.txt
for E in 0 1 2; do
  DM_GEMM_DMA_EARLY=$E timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-h2d-leg --prof-steps 1 > $O/bench_early$E.json 2> $O/bench_early$E.err
  python - <<PY
import json
d=json.load(open('$O/bench_early$E.json')); print('EARLY=$E ms/step', d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_gemm']['ms_per_step'], d['roofline']['traffic'])
PY
done
bash scripts/gemm_sq_counters.sh r05 | tail -8
