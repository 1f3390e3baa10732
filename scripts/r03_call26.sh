#!/bin/bash
O=gpurun_out/r03z; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_training_step.py -m gpu -q -x -k "twins or amp or bf16" > $O/pt.log 2>&1; tail -4 $O/pt.log
