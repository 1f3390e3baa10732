#!/bin/bash
# round 3, GPU call 2: software-pipelined bf16-pipe GEMM kernel (split fp32 products and bf16 operands)
set -x
O=gpurun_out/r03b; mkdir -p $O
export PYTHONPATH=$PWD
( cd scripts && python gemm_modes.py > ../$O/gemm_split_pipe.jsonl 2> ../$O/gemm_split_pipe.err )
( cd scripts && python gemm_modes.py --bf16 > ../$O/gemm_bf16_pipe.jsonl 2> ../$O/gemm_bf16_pipe.err )
( cd scripts && python gemm_modes.py --dist positive --only 0,3,6,9,19,22,26,27 > ../$O/gemm_split_pipe_pos.jsonl 2>> ../$O/gemm_split_pipe.err )
DM_CHAIN_GRAPH=0 timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
DM_CHAIN_GRAPH=0 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --shape-table $O/shapes_split_pipe.txt > $O/bench_split_pipe.json 2> $O/bench_split_pipe.err
DM_CHAIN_GRAPH=0 DM_GEMM_NO_PIPE=1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_split_nopipe.json 2> $O/bench_split_nopipe.err
DM_CHAIN_GRAPH=0 DM_PANEL_MIN_ROWS=1000000000 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_split_pipe_nopanel.json 2> $O/bench_split_pipe_nopanel.err
DM_CHAIN_GRAPH=0 DM_PANEL_MIN_ROWS=1000000000 DM_MLP_NO_CHAIN=1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_split_pipe_nopanel_nochain.json 2> $O/bench_split_pipe_nopanel_nochain.err
DM_CHAIN_GRAPH=0 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --dtype bf16 --shape-table $O/shapes_bf16_pipe.txt > $O/bench_bf16_pipe.json 2> $O/bench_bf16_pipe.err
DM_CHAIN_GRAPH=0 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --emulate-world 8 --prof-steps 0 > $O/shard_split_pipe.json 2> $O/shard_split_pipe.err
for f in $O/bench_*.json $O/shard_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], 'ms', round(d['ms_per_step'],2), 'host', round(d['host_enqueue_ms_per_step'],2), 'gemm', d['roofline'] and (round(d['roofline']['all_gemm']['tflops'],1), round(d['roofline']['all_gemm']['ms_per_step'],2)))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
python - <<'PY'
import json
for f in ('gpurun_out/r03b/gemm_split_pipe.jsonl','gpurun_out/r03b/gemm_bf16_pipe.jsonl'):
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); print(d['mode'], d['what'][:28].ljust(28), round(d['tflops'],1), d['err'] and '%.2e'%d['err']['rel_l2'])
PY
