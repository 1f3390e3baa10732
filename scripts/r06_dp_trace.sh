#!/bin/bash
# kernel traces of the 25-column shard step with and without the (one-rank) data-parallel path: where the +1.1 ms goes
set -u
OUT=$PWD/gpurun_out/r06; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
B="--reps 1 --steps 8 --warmup 4 --no-cpu-baseline --no-h2d-leg --prof-steps 0 --dtype f32 --emulate-world 2"
for V in none late native; do
  FD="--force-dp"; E="DM_X=0"
  [ $V = none ] && FD=""
  [ $V = native ] && E="DM_DP_NATIVE=1"
  env $E timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/dpt_$V -o t -- python $REPO/bench.py $B $FD > $OUT/dpt_$V.json 2> $OUT/dpt_$V.err
  echo "=== $V"; python $REPO/scripts/r06_dp_trace.py $(find /tmp/dpt_$V -name "*kernel_trace.csv" | head -1)
done > $OUT/r06_dp_trace.txt 2>&1
cat $OUT/r06_dp_trace.txt
