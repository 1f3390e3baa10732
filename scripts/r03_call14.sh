#!/bin/bash
O=gpurun_out/r03n; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "bf16" > $O/pt1.log 2>&1; tail -15 $O/pt1.log
timeout 900 python -m pytest tests/test_gpu_training_step.py -m gpu -q -x -k "twins or amp or bf16 or conv" > $O/pt2.log 2>&1; tail -25 $O/pt2.log
timeout 300 python bench.py --dtype bf16 --no-h2d-leg --no-cpu-baseline --steps 30 > $O/bench_bf16.json 2> $O/bench_bf16.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03n/bench_bf16.json').read().strip().splitlines()[-1])
print('bf16', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['roofline']['all_gemm'])
PY
DM_BF16_NO_TWINS=1 timeout 300 python bench.py --dtype bf16 --no-h2d-leg --no-cpu-baseline --steps 30 > $O/bench_bf16_off.json 2> $O/bench_bf16_off.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03n/bench_bf16_off.json').read().strip().splitlines()[-1])
print('bf16 twins off', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['roofline']['all_gemm'])
PY
