#!/bin/bash
set -x
O=gpurun_out/r03i; mkdir -p $O
export PYTHONPATH=$PWD
timeout 1200 python -m pytest tests -m gpu -q -x -k "conv or uint8 or tiny_two_steps or reference_goldens or atari_literal or device_ring" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-h2d-leg > $O/bench_direct.json 2> $O/bench_direct.err
DM_CONV_NO_DIRECT=1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-h2d-leg > $O/bench_nodirect.json 2> $O/bench_nodirect.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-h2d-leg --prof-steps 0 --no-overlap > $GRAFT_REPO_ROOT/$O/ks_serial_bench.json 2> $GRAFT_REPO_ROOT/$O/ks_serial.err
cp $(find /tmp/prof_ks -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/kernel_stats_serial.csv
cd $GRAFT_REPO_ROOT
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], 'ms', round(d['ms_per_step'],2), 'host', round(d['host_enqueue_ms_per_step'],2), 'gemm ms', d['roofline'] and round(d['roofline']['all_gemm']['ms_per_step'],2))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r03i/kernel_stats_serial.csv')):
    if any(k in r['Name'] for k in ('enc_l1','dec_l4','im2col','col2im','mse_image')):
        print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
