#!/bin/bash
O=gpurun_out/r03r; mkdir -p $O
timeout 120 scripts/microbench/xcd_barrier 32 2000 4096 23 512 3 > $O/xcd_nf_32.txt 2>&1; grep -v placement $O/xcd_nf_32.txt | head -4
timeout 120 scripts/microbench/xcd_barrier 16 2000 4096 23 512 3 > $O/xcd_nf_16.txt 2>&1; grep -v placement $O/xcd_nf_16.txt | head -4
timeout 120 scripts/microbench/xcd_barrier 32 2000 4096 23 64 3 > $O/xcd_nf_32_64.txt 2>&1; grep -v placement $O/xcd_nf_32_64.txt | head -4
