#!/bin/bash
O=gpurun_out/r03u; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python -m pytest tests/test_gpu_training_step.py -m gpu -q -x -k "persistent or rssm_sequence or goldens" > $O/pt.log 2>&1; tail -4 $O/pt.log
timeout 200 python scripts/persist_prof.py 50 > $O/prof50.txt 2>&1; grep "persist=0" $O/prof50.txt
timeout 200 python scripts/persist_prof.py 7 > $O/prof7.txt 2>&1; grep "persist=0" $O/prof7.txt
timeout 200 python bench.py --no-h2d-leg --no-cpu-baseline --steps 30 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03u/bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'])
PY
