#!/bin/bash
O=gpurun_out/r03m; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python scripts/gemm_h_bench.py > $O/h_default.txt 2>&1; cut -c1-215 $O/h_default.txt
DM_GEMM_H_NT=1 timeout 600 python scripts/gemm_h_bench.py > $O/h_nt.txt 2>&1; echo "== NT loads"; cut -c1-215 $O/h_nt.txt
