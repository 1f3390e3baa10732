#!/bin/bash
# one GPU call at the end of a round: the core profile set (kernel statistics, counter passes, the bench line) first, then the
# whole -m gpu suite, then the other configurations; "lab" (tile-kernel laboratory, SQ counters, tile sweep) only when asked:
#   gpurun --timeout 3000 -- 'bash scripts/gpu_round_end.sh r05 [lab]'
TAG=${1:-r06}
mkdir -p gpurun_out/$TAG
PARTS=core bash scripts/collect_profiles.sh $TAG > gpurun_out/collect_${TAG}_core.log 2>&1
python -c "import json; d=json.load(open('gpurun_out/$TAG/${TAG}_bench.json')); print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'])"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/$TAG/${TAG}_gputest.txt 2>&1; echo "rc $?" >> gpurun_out/$TAG/${TAG}_gputest.txt
tail -3 gpurun_out/$TAG/${TAG}_gputest.txt
PARTS="configs ${2:-}" bash scripts/collect_profiles.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
tail -14 gpurun_out/collect_$TAG.log
