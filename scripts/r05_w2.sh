#!/bin/bash
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 --no-h2d-leg --pmc-json /nonexistent --pipeline --emulate-world $1 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$2 world $1:', round(d['ms_per_step'],2))"; }
run 2 default
DM_RSSM_LDS_COOP=0 run 2 coop0
DM_GEMM_DMA=0 run 2 dma0
DM_RSSM_LDS=0 run 2 lds0
DM_RSSM_LDS_COOP=0 run 4 coop0
run 4 default
DM_RSSM_LDS_COOP=0 run 8 coop0
run 8 default
