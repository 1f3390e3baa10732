#!/bin/bash
set -x
O=gpurun_out/r03d; mkdir -p $O
export PYTHONPATH=$PWD
python scripts/chain_graph_bench.py 50 50 15 > $O/chain_graph_b50.json 2> $O/chain_graph_b50.err
python scripts/chain_graph_bench.py 7 50 15 > $O/chain_graph_b7.json 2> $O/chain_graph_b7.err
cat $O/chain_graph_b50.json $O/chain_graph_b7.json
for only in rssm_sequence_fwd,dream_rollout rssm_sequence_bwd dream_rollout rssm_sequence_fwd; do
  tag=$(echo $only | tr ',' '+')
  DM_CHAIN_GRAPH_ONLY=$only python bench.py --steps 30 --warmup 10 --no-cpu-baseline --emulate-world 8 --prof-steps 0 > $O/shard_f32_only_$tag.json 2> $O/shard_f32_only_$tag.err
  DM_CHAIN_GRAPH_ONLY=$only python bench.py --steps 30 --warmup 10 --no-cpu-baseline --prof-steps 0 --dtype bf16 > $O/bench_bf16_only_$tag.json 2> $O/bench_bf16_only_$tag.err
done
DM_CHAIN_GRAPH=0 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --emulate-world 8 --prof-steps 0 > $O/shard_f32_nograph.json 2> $O/shard_f32_nograph.err
DM_CHAIN_GRAPH=0 timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_nograph.log 2>&1; echo "pytest rc $?" >> $O/pytest_nograph.log
timeout 900 python -m pytest tests/test_gpu_training_step.py -m gpu -q -x -k "chain_graphs or overwritten or two_steps" > $O/pytest_graph.log 2>&1; echo "pytest rc $?" >> $O/pytest_graph.log
tail -5 $O/pytest_nograph.log $O/pytest_graph.log
for f in $O/bench_*.json $O/shard_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], 'ms', round(d['ms_per_step'],2), 'host', round(d['host_enqueue_ms_per_step'],2), d.get('chain_graphs'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
