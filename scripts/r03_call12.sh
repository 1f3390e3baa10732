#!/bin/bash
O=gpurun_out/r03l; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python scripts/gemm_h_bench.py > $O/h_default.txt 2>&1; cat $O/h_default.txt
for t in 1 2 3; do DM_GEMM_TILE=$t timeout 300 python scripts/gemm_h_bench.py > $O/h_tile$t.txt 2>&1; echo "== tile $t"; cut -c1-200 $O/h_tile$t.txt; done
