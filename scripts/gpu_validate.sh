#!/bin/bash
# One gpurun call's worth of validation, run on the GPU box from the repo root:
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash scripts/gpu_validate.sh'
# GPU test suite, the driver's smoke(), the default bench line; outputs under gpurun_out/validate/.
O=gpurun_out/validate; mkdir -p $O
export PYTHONPATH=$PWD
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
