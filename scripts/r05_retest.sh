#!/bin/bash
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r05/r05_gputest.txt 2>&1; echo "rc $?" >> gpurun_out/r05/r05_gputest.txt
tail -3 gpurun_out/r05/r05_gputest.txt
timeout 600 python -m pytest tests -m gpu -s -q -k "atari_literal" 2>&1 | grep -i "imagin\|trajector\|posterior" 
python scripts/chain_fit.py > gpurun_out/r05/chain_fit.txt 2>&1; cat gpurun_out/r05/chain_fit.txt
