#!/bin/bash
# stream-priority A/B on the default bench (diagnostic)
O=gpurun_out/prio; mkdir -p $O
export PYTHONPATH=$PWD
run() { # name, env...
  n=$1; shift
  for dt in f32 bf16; do
    env "$@" timeout 200 python bench.py --dtype $dt --no-h2d-leg --no-cpu-baseline --steps 30 --pmc-json /nonexistent 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', '$dt', round(d['ms_per_step'],3))"
  done
}
run base X=1
run wm0 DM_WM_PRIO=0
run main_hi_wm0 DM_MAIN_PRIO=-1 DM_WM_PRIO=0
run side0 DM_WGRAD_SIDE_PRIO=0
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 --emulate-world 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shard', round(d['ms_per_step'],3))"
