#!/bin/bash
# kernel trace of the bench workload -> per-queue timeline: bash scripts/gpu_trace.sh <dtype> [tag] [extra bench.py arguments, e.g. "--emulate-world 8 --pipeline"] [suffix]
DT=${1:-f32}; TAG=${2:-trace}; EXTRA=${3:-}; SUF=${4:-}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$PWD
REPO=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tq_$DT$SUF -o t -- python $REPO/bench.py --reps 1 --dtype $DT $EXTRA --steps 6 --warmup 3 --no-cpu-baseline --no-h2d-leg --prof-steps 0 --pmc-json /nonexistent > $O/bench_$DT.json 2> $O/err_$DT.txt
python $REPO/scripts/trace_queues.py $(find /tmp/prof_tq_$DT$SUF -name "*kernel_trace.csv" | head -1) 2 ${BIN_MS:-0.5} > $O/queues_$DT$SUF.txt 2>&1
head -8 $O/queues_$DT$SUF.txt
