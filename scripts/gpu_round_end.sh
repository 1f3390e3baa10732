#!/bin/bash
# one GPU call at the end of a round: the whole -m gpu suite, then the profile collection
#   gpurun --timeout 4500 -- 'bash scripts/gpu_round_end.sh r05'
TAG=${1:-r05}
mkdir -p gpurun_out/$TAG
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/$TAG/${TAG}_gputest.txt 2>&1; echo "rc $?" >> gpurun_out/$TAG/${TAG}_gputest.txt
tail -3 gpurun_out/$TAG/${TAG}_gputest.txt
bash scripts/collect_profiles.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
tail -14 gpurun_out/collect_$TAG.log
