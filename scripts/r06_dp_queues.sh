#!/bin/bash
# which hardware queues the step's streams land on, with and without communicators in the process (25-column shard, kernel traces)
set -u
OUT=$PWD/gpurun_out/r06; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
B="--reps 1 --steps 8 --warmup 4 --no-cpu-baseline --no-h2d-leg --prof-steps 0 --dtype f32 --emulate-world 2"
i=0
run() {   # label, env..., --, bench flags
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  i=$((i+1))
  env "${envs[@]}" timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/dpq_$i -o t -- python $REPO/bench.py $B "$@" > $OUT/dpq_$i.json 2> $OUT/dpq_$i.err
  echo "=== $label: bench line $(python -c "import json,sys; print(round(json.loads([l for l in open('$OUT/dpq_$i.json') if l.startswith('{')][-1])['ms_per_step'],3))" 2>/dev/null) ms (under the profiler)"
  python $REPO/scripts/r06_dp_trace.py $(find /tmp/dpq_$i -name "*kernel_trace.csv" | head -1) 0
}
{
run "no DP" DM_X=0 --
run "no DP, 8 hw queues" GPU_MAX_HW_QUEUES=8 --
run "late torch" DM_X=0 -- --force-dp
run "late torch, 8 hw queues" GPU_MAX_HW_QUEUES=8 -- --force-dp
run "late torch, 2 hw queues" GPU_MAX_HW_QUEUES=2 -- --force-dp
run "native late, gloo control plane" DM_DP_NATIVE=1 DM_BENCH_FORCE_BACKEND=gloo -- --force-dp
run "idle group created before the model" DM_BENCH_DP_IDLE=1 -- --force-dp
} > $OUT/r06_dp_queues.txt 2>&1
cat $OUT/r06_dp_queues.txt
