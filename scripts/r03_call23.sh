#!/bin/bash
O=gpurun_out/r03w; mkdir -p $O
export PYTHONPATH=$PWD
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/r03_bench.json 2> $O/bench.err; tail -c 400 $O/r03_bench.json
