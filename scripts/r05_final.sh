#!/bin/bash
# final round-5 GPU call: the whole -m gpu suite, then the profile collection
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r05/r05_gputest.txt 2>&1; echo "rc $?" >> gpurun_out/r05/r05_gputest.txt
tail -3 gpurun_out/r05/r05_gputest.txt
bash scripts/collect_profiles.sh r05 > gpurun_out/collect_r05.log 2>&1
tail -14 gpurun_out/collect_r05.log
