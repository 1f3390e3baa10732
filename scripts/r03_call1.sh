#!/bin/bash
# round 3, GPU call 1: split-bf16 fp32 products and linear chain graphs - parity, error vs fp64, speed
set -x
O=gpurun_out/r03; mkdir -p $O
export PYTHONPATH=$PWD
( cd scripts && python gemm_modes.py > ../$O/gemm_split.jsonl 2> ../$O/gemm_split.err )
( cd scripts && DM_FP32_NATIVE=1 python gemm_modes.py > ../$O/gemm_native.jsonl 2> ../$O/gemm_native.err )
( cd scripts && python gemm_modes.py --bf16 > ../$O/gemm_bf16.jsonl 2> ../$O/gemm_bf16.err )
( cd scripts && python gemm_modes.py --dist positive --only 0,3,6,9,19,22,26,27 > ../$O/gemm_split_pos.jsonl 2>> ../$O/gemm_split.err )
( cd scripts && DM_FP32_NATIVE=1 python gemm_modes.py --dist positive --only 0,3,6,9,19,22,26,27 > ../$O/gemm_native_pos.jsonl 2>> ../$O/gemm_native.err )
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_split_graph.log 2>&1; echo "pytest rc $?" >> $O/pytest_split_graph.log
if grep -q "failed" $O/pytest_split_graph.log; then
  DM_CHAIN_GRAPH=0 timeout 900 python -m pytest tests -m gpu -q > $O/pytest_split_nograph.log 2>&1
  DM_FP32_NATIVE=1 timeout 900 python -m pytest tests -m gpu -q > $O/pytest_native_graph.log 2>&1
fi
python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_split_graph.json 2> $O/bench_split_graph.err
DM_CHAIN_GRAPH_DEBUG=1 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --prof-steps 0 > /dev/null 2> $O/graph_debug.err
DM_CHAIN_GRAPH=0 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_split_nograph.json 2> $O/bench_split_nograph.err
DM_FP32_NATIVE=1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_native_graph.json 2> $O/bench_native_graph.err
DM_FP32_NATIVE=1 DM_CHAIN_GRAPH=0 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_native_nograph.json 2> $O/bench_native_nograph.err
DM_PANEL_MIN_ROWS=1000000000 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_split_nopanel.json 2> $O/bench_split_nopanel.err
DM_PANEL_MIN_ROWS=1000000000 DM_MLP_NO_CHAIN=1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_split_nopanel_nochain.json 2> $O/bench_split_nopanel_nochain.err
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --emulate-world 8 --prof-steps 0 > $O/shard_split_graph.json 2> $O/shard_split_graph.err
DM_CHAIN_GRAPH=0 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --emulate-world 8 --prof-steps 0 > $O/shard_split_nograph.json 2> $O/shard_split_nograph.err
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --dtype bf16 > $O/bench_bf16_graph.json 2> $O/bench_bf16_graph.err
DM_CHAIN_GRAPH=0 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --dtype bf16 > $O/bench_bf16_nograph.json 2> $O/bench_bf16_nograph.err
tail -3 $O/pytest_split_graph.log
for f in $O/bench_*.json $O/shard_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], 'ms', round(d['ms_per_step'],2), 'host', round(d['host_enqueue_ms_per_step'],2), 'gemm', d['roofline'] and round(d['roofline']['all_gemm']['tflops'],1), d.get('chain_graphs'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
