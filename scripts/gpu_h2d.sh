#!/bin/bash
# H2D-included leg with / without the weight-gradient side stream (diagnostic)
export PYTHONPATH=$PWD
for v in 1 0; do
  DM_WGRAD_SIDE=$v timeout 300 python bench.py --no-cpu-baseline --steps 20 --prof-steps 0 --pmc-json /nonexistent 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side=$v', round(d['ms_per_step'],3), 'h2d', round(d['h2d_included']['ms_per_step'],3))"
done
