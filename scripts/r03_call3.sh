#!/bin/bash
# round 3, GPU call 3: chain graphs with the step arena (stable pointers)
set -x
O=gpurun_out/r03c; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
for mode in graph nograph; do
  if [ $mode = nograph ]; then export DM_CHAIN_GRAPH=0; else unset DM_CHAIN_GRAPH; fi
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_f32_$mode.json 2> $O/bench_f32_$mode.err
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline --dtype bf16 > $O/bench_bf16_$mode.json 2> $O/bench_bf16_$mode.err
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline --emulate-world 8 --prof-steps 0 > $O/shard_f32_$mode.json 2> $O/shard_f32_$mode.err
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline --emulate-world 8 --prof-steps 0 --dtype bf16 > $O/shard_bf16_$mode.json 2> $O/shard_bf16_$mode.err
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline --emulate-world 8 --prof-steps 0 --no-overlap > $O/shard_f32_${mode}_nooverlap.json 2> $O/shard_f32_${mode}_nooverlap.err
done
unset DM_CHAIN_GRAPH
DM_CHAIN_GRAPH_DEBUG=1 python bench.py --steps 6 --warmup 4 --no-cpu-baseline --prof-steps 0 > /dev/null 2> $O/graph_debug.err
for f in $O/bench_*.json $O/shard_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], 'ms', round(d['ms_per_step'],2), 'host', round(d['host_enqueue_ms_per_step'],2), d.get('chain_graphs'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
grep -c miss $O/graph_debug.err; head -12 $O/graph_debug.err
