#!/bin/bash
O=gpurun_out/r03o; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_training_step.py -m gpu -q -k "twins" > $O/pt2.log 2>&1; tail -12 $O/pt2.log
timeout 300 python bench.py --dtype bf16 --no-h2d-leg --no-cpu-baseline --steps 20 --shape-table $O/shapes_on.txt > $O/bench_bf16.json 2> $O/bench_bf16.err
DM_BF16_NO_TWINS=1 timeout 300 python bench.py --dtype bf16 --no-h2d-leg --no-cpu-baseline --steps 20 --shape-table $O/shapes_off.txt > $O/bench_bf16_off.json 2> $O/bench_bf16_off.err
head -50 $O/shapes_on.txt
