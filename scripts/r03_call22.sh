#!/bin/bash
set -x
O=gpurun_out/r03v; mkdir -p $O
export PYTHONPATH=$PWD
bash scripts/collect_profiles.sh r03 > $O/collect.log 2>&1
tail -14 $O/collect.log
