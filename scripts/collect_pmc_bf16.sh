#!/bin/bash
# HBM traffic of the bf16 step (two rocprofv3 PMC passes, as collect_profiles.sh does for fp32) -> <tag>_pmc_traffic_bf16.json,
# then the bf16 bench line reading it.  Run on the GPU box from the repo root: bash scripts/collect_pmc_bf16.sh r03
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
SHA=$(python bench.py --csrc-sha)
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/profh_$C -o p -- python $REPO/bench.py --reps 1 --dtype bf16 --steps 2 --warmup 1 --no-overlap --no-cpu-baseline --no-h2d-leg --prof-steps 0 > $OUT/pmch_$C.json 2> $OUT/pmch_$C.err
done
python $REPO/scripts/pmc_traffic.py $(find /tmp/profh_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/profh_WRITE_SIZE -name "*counter_collection.csv" | head -1) 3 $SHA > $OUT/${TAG}_pmc_traffic_bf16.json
cd $REPO
cp $OUT/${TAG}_pmc_traffic_bf16.json profiles/${TAG}_pmc_traffic_bf16.json
python bench.py --dtype bf16 --no-cpu-baseline --shape-table $OUT/${TAG}_gemm_shapes_bf16.txt > $OUT/${TAG}_bench_bf16.json 2>/dev/null
python - << PY
import json
p = json.load(open('$OUT/${TAG}_pmc_traffic_bf16.json'))
print('bf16 HBM GB/step', p['total_gb_per_step'])
b = json.loads(open('$OUT/${TAG}_bench_bf16.json').read().strip().splitlines()[-1])
print('bf16:', b['ms_per_step'], b['value'], b['roofline']['traffic'])
PY
