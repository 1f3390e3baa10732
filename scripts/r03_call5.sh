#!/bin/bash
set -x
O=gpurun_out/r03e; mkdir -p $O
export PYTHONPATH=$PWD
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
python bench.py --steps 30 --warmup 10 > $O/bench_f32.json 2> $O/bench_f32.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload atari-native > $O/bench_native_workload.json 2> $O/bench_native_workload.err
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --workload dmc --dtype bf16 > $O/bench_dmc_bf16.json 2> $O/bench_dmc_bf16.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], 'ms', round(d['ms_per_step'],2), 'host', round(d['host_enqueue_ms_per_step'],2), 'h2d', d.get('h2d_included') and round(d['h2d_included']['ms_per_step'],2), 'cpu', d.get('cpu_baseline') and d['cpu_baseline'].get('value'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
tail -3 $O/*.err
