#!/bin/bash
O=gpurun_out/r03p; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_training_step.py -m gpu -q -k "twins or amp or bf16 or dream" > $O/pt2.log 2>&1; tail -12 $O/pt2.log
timeout 300 python bench.py --dtype bf16 --no-h2d-leg --no-cpu-baseline --steps 30 --shape-table $O/shapes_on.txt > $O/bench_bf16.json 2> $O/bench_bf16.err
DM_BF16_NO_TWINS=1 timeout 300 python bench.py --dtype bf16 --no-h2d-leg --no-cpu-baseline --steps 30 --shape-table $O/shapes_off.txt > $O/bench_bf16_off.json 2> $O/bench_bf16_off.err
timeout 300 python bench.py --dtype bf16 --no-h2d-leg --no-cpu-baseline --steps 30 > $O/bench_bf16_b.json 2> $O/bench_bf16_b.err
DM_BF16_NO_TWINS=1 timeout 300 python bench.py --dtype bf16 --no-h2d-leg --no-cpu-baseline --steps 30 > $O/bench_bf16_off_b.json 2> $O/bench_bf16_off_b.err
python scripts/compare_shape_tables.py $O/shapes_on.txt $O/shapes_off.txt $O/bench_bf16.json $O/bench_bf16_off.json $O/bench_bf16_b.json $O/bench_bf16_off_b.json
