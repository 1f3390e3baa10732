#!/bin/bash
mkdir -p gpurun_out/r05
O=gpurun_out/r05
for C in 1 2 3 5 1 3; do
  DM_PIPELINE_CHUNKS=$C timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-h2d-leg --prof-steps 0 --pmc-json /nonexistent > $O/bench_chunks$C.json 2> $O/bench_chunks$C.err
  python - <<PY
import json
d=json.load(open('$O/bench_chunks$C.json')); print('CHUNKS=$C ms/step', d['ms_per_step'], 'loss', d['loss_model_last'])
PY
done
for C in 1 3; do
  DM_PIPELINE_CHUNKS=$C timeout 600 python bench.py --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-h2d-leg --prof-steps 0 --pmc-json /nonexistent > $O/bench_chunks_bf16_$C.json 2> /dev/null
  python - <<PY
import json
d=json.load(open('$O/bench_chunks_bf16_$C.json')); print('bf16 CHUNKS=$C ms/step', d['ms_per_step'])
PY
done
