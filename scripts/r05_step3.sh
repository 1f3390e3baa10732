#!/bin/bash
mkdir -p gpurun_out/r05
python scripts/gemm_tile_sweep.py > gpurun_out/r05/tile_sweep.txt 2>&1
tail -3 gpurun_out/r05/tile_sweep.txt
