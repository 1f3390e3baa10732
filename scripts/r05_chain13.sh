#!/bin/bash
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_training_step.py -x -q -k "mlp_head or dream_rollout" > $O/t_chain.txt 2>&1; echo "rc $?" >> $O/t_chain.txt
tail -3 $O/t_chain.txt
for W in 13 4 13 4; do
  DM_CHAIN_WAVES=$W timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-h2d-leg --prof-steps 1 --pmc-json /nonexistent > $O/bench_cw_$W.json 2> $O/bench_cw_$W.err
  python - <<PY
import json
d=json.load(open('$O/bench_cw_$W.json')); k={x['kernel']:(round(x['ms_per_step'],2),round(x['avg_launch_us'],1)) for x in d['roofline']['kinds'] if 'chain' in x['kernel']}
print('CHAIN_WAVES=$W ms/step', d['ms_per_step'], d['roofline']['all_gemm']['ms_per_step'], k)
PY
done
for W in 13 4; do
  for N in 8 4; do
  DM_CHAIN_WAVES=$W python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 --no-h2d-leg --pmc-json /nonexistent --pipeline --emulate-world $N 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('CHAIN_WAVES=$W world $N:', round(d['ms_per_step'],2))"
  DM_CHAIN_WAVES=$W DM_CHAIN_MIN_ROWS=256 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 --no-h2d-leg --pmc-json /nonexistent --pipeline --emulate-world $N 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('CHAIN_WAVES=$W MIN_ROWS=256 world $N:', round(d['ms_per_step'],2))"
  done
done
DM_CHAIN_WAVES=13 python bench.py --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline --prof-steps 0 --no-h2d-leg --pmc-json /nonexistent 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('bf16 W13', round(d['ms_per_step'],2))"
DM_CHAIN_WAVES=4 python bench.py --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline --prof-steps 0 --no-h2d-leg --pmc-json /nonexistent 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('bf16 W4', round(d['ms_per_step'],2))"
