#!/bin/bash
O=gpurun_out/ab2; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "gemm" > $O/pt.log 2>&1; tail -3 $O/pt.log
for v in 1 0 2; do
  DM_GEMM_PIPE32=$v timeout 200 python bench.py --no-h2d-leg --no-cpu-baseline --steps 40 --shape-table $O/shapes_$v.txt > $O/bench_$v.json 2> $O/bench_$v.err
done
python - <<'PY'
import json
for v in (1,0,2):
    d=json.loads(open(f'gpurun_out/ab2/bench_{v}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('PIPE32=%d' % v, round(d['value'],3), round(d['ms_per_step'],3), d['loss_model_last'], 'dominant', r['kernel'], round(r['frac'],3), 'all_gemm ms', round(r['all_gemm']['ms_per_step'],2))
PY
