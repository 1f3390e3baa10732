#!/bin/bash
# H2D-included leg of the bf16 step, repeated in fresh processes (diagnostic)
export PYTHONPATH=$PWD
p() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'h2d', round(d['h2d_included']['ms_per_step'],3))"; }
timeout 100 python bench.py --dtype bf16 --no-cpu-baseline --prof-steps 0 --steps 20 2>/dev/null | p bf16_a
timeout 100 python bench.py --dtype bf16 --no-cpu-baseline --prof-steps 0 --steps 20 2>/dev/null | p bf16_b
DM_RING_NO_PREFETCH=1 timeout 100 python bench.py --dtype bf16 --no-cpu-baseline --prof-steps 0 --steps 20 2>/dev/null | p bf16_noprefetch
