#!/bin/bash
# /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash scripts/gpu_collect.sh'   then copy gpurun_out/r03/r03_* into profiles/
export PYTHONPATH=$PWD
bash scripts/collect_profiles.sh r03 > gpurun_out/collect.log 2>&1
tail -12 gpurun_out/collect.log; tail -3 gpurun_out/r03/collect_bf16.log
