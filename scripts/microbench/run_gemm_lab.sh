#!/bin/bash
# gpurun -- bash scripts/microbench/run_gemm_lab.sh   (results -> gpurun_out/gemm_lab.txt)
mkdir -p gpurun_out
rocm-smi --showclocks 2>/dev/null | head -20 > gpurun_out/gemm_lab_smi.txt
timeout 900 scripts/microbench/gemm_lab ${LAB_REPS:-20} pydreamer_amd/libdreamer_hip.so > gpurun_out/gemm_lab.txt 2>&1
echo "exit $?" >> gpurun_out/gemm_lab.txt
tail -5 gpurun_out/gemm_lab.txt
