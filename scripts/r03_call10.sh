#!/bin/bash
set -x
O=gpurun_out/r03j; mkdir -p $O
export PYTHONPATH=$PWD
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
bash scripts/collect_profiles.sh r03 > $O/collect.log 2>&1
tail -12 $O/collect.log
