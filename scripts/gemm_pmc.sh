#!/bin/bash
# SQ counters of dm_gemm_f32 on a few of the step's shapes (diagnostic): where the waves' cycles go.
# bash scripts/gemm_pmc.sh "9,10,0" [DM_GEMM_TILE value]
export TMPDIR=/tmp
REPO=$PWD
O=$REPO/gpurun_out/gemm_pmc; mkdir -p $O
SHAPES=${1:-9,10,0}
if [ -n "${2:-}" ]; then export DM_GEMM_TILE=$2; fi
cd /tmp
rm -rf /tmp/prof_gemm
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/prof_gemm -o g -- python $REPO/scripts/gemm_modes.py --only $SHAPES --reps 4 > $O/run_${2:-auto}.txt 2> $O/run.err
python - << PY
import csv, glob, collections
f = glob.glob('/tmp/prof_gemm/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'gemm' not in k: continue
    key = (k[:70], r.get('Grid_Size'))
    acc[key][r['Counter_Name']] += float(r['Counter_Value'])
for key, c in acc.items():
    wc = c.get('SQ_WAVE_CYCLES', 1) or 1
    print(key)
    print('   ', {k: round(v / wc, 3) for k, v in c.items() if k != 'SQ_WAVE_CYCLES'}, 'wave_cycles', int(wc))
PY
