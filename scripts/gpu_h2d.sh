#!/bin/bash
# H2D-included leg, repeated in fresh processes (diagnostic: it was bimodal with a copy stream of its own)
export PYTHONPATH=$PWD
p() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'h2d', round(d['h2d_included']['ms_per_step'],3))"; }
python -m pytest tests/test_gpu_replay.py -q -m gpu 2>&1 | tail -1
for i in 1 2 3 4; do timeout 300 python bench.py --no-cpu-baseline --prof-steps 0 --steps 20 2>/dev/null | p run$i; done
