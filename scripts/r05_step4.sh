#!/bin/bash
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_primitives.py -x -q -k "gemm" > $O/t_gemm.txt 2>&1; echo "gemm rc $?" >> $O/t_gemm.txt
tail -2 $O/t_gemm.txt
python scripts/gemm_tile_sweep.py > $O/tile_sweep2.txt 2>&1
head -12 $O/tile_sweep2.txt; tail -1 $O/tile_sweep2.txt
for T in 1 0; do
  DM_GEMM_T160=$T timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-h2d-leg --pmc-json /nonexistent --shape-table $O/shapes_t160_$T.txt > $O/bench_t160_$T.json 2> $O/bench_t160_$T.err
  python - <<PY
import json
d=json.load(open('$O/bench_t160_$T.json')); print('T160=$T ms/step', d['ms_per_step'], 'dom', d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('all_gemm'))
PY
done
