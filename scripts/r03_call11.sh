#!/bin/bash
set -x
O=gpurun_out/r03k; mkdir -p $O
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -8 $O/pytest.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
timeout 300 python bench.py --no-h2d-leg --steps 20 --warmup 5 --emulate-world 8 --no-cpu-baseline > $O/bench_shard.json 2> $O/bench_shard.err; tail -c 600 $O/bench_shard.json
