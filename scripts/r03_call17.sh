#!/bin/bash
O=gpurun_out/r03q; mkdir -p $O
export PYTHONPATH=$PWD
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -8 $O/pytest.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
