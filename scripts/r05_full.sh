#!/bin/bash
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 3000 python -m pytest tests -m gpu -q -x > $O/t_all.txt 2>&1; echo "all rc $?" >> $O/t_all.txt
tail -5 $O/t_all.txt
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-h2d-leg --pmc-json /nonexistent --shape-table $O/shapes_now.txt > $O/bench_now.json 2> $O/bench_now.err
python - <<PY
import json
d=json.load(open('$O/bench_now.json')); print('ms/step', d['ms_per_step'], 'dom', d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('all_gemm'))
PY
