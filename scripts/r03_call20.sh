#!/bin/bash
O=gpurun_out/r03t; mkdir -p $O
export PYTHONPATH=$PWD
timeout 200 python scripts/persist_prof.py 50 > $O/prof50.txt 2>&1; tail -6 $O/prof50.txt
timeout 200 python scripts/persist_prof.py 7 > $O/prof7.txt 2>&1; tail -6 $O/prof7.txt
