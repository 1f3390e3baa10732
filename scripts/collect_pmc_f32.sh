#!/bin/bash
# HBM traffic + per-kernel durations of the fp32 step (two rocprofv3 PMC passes), then the default bench line reading them: the
# fp32 half of collect_profiles.sh on its own.  Run on the GPU box from the repo root: bash scripts/collect_pmc_f32.sh
set -u
TAG=r04; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD; SHA=$(python bench.py --csrc-sha)
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -o p -- python $REPO/bench.py --reps 1 --steps 2 --warmup 1 --no-overlap --no-cpu-baseline --no-h2d-leg --prof-steps 0 > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
python $REPO/scripts/pmc_traffic.py $(find /tmp/prof_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/prof_WRITE_SIZE -name "*counter_collection.csv" | head -1) 3 $SHA > $OUT/${TAG}_pmc_traffic.json
cd $REPO
cp $OUT/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
timeout 300 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
tail -c 300 $OUT/${TAG}_bench.json
