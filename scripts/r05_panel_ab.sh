#!/bin/bash
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_primitives.py -x -q -k "gemm" 2>&1 | tail -1
for P in 16384 100000000 16384 100000000; do
  DM_PANEL_MIN_ROWS=$P timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-h2d-leg --prof-steps 1 --pmc-json /nonexistent > $O/bench_panel$P.json 2> $O/bench_panel$P.err
  python - <<PY
import json
d=json.load(open('$O/bench_panel$P.json')); print('PANEL_MIN_ROWS=$P ms/step', d['ms_per_step'], d['roofline']['all_gemm']['ms_per_step'])
PY
done
