#!/bin/bash
# Repeat the GPU suite (flakiness check): bash scripts/gpu_flaky.sh 3
O=gpurun_out/flaky; mkdir -p $O
export PYTHONPATH=$PWD
for i in $(seq 1 ${1:-2}); do
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/run$i.log 2>&1; tail -1 $O/run$i.log
done
