#!/bin/bash
# SQ counters of the fp32 tile kernels (LDS-DMA loop on / off) on 4096^3 and the rollout's 2500 x 1800 x 1000 gate product:
#   bash scripts/gemm_sq_counters.sh <tag>   -> gpurun_out/<tag>/<tag>_gemm_sq_counters.txt
# Counters only (no tracing domains besides --kernel-trace); one pass of 8 SQ counters + one of GRBM_GUI_ACTIVE (effective clock).
set -u
TAG=${1:-r05}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for D in 1 0; do
  DM_GEMM_DMA=$D rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES \
    --kernel-trace --output-format csv -d /tmp/prof_sq_$D -o p -- python $REPO/scripts/gemm_bench.py --only 0,9 --reps 4 > $OUT/sq_$D.log 2>&1
  DM_GEMM_DMA=$D rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/prof_grbm_$D -o p -- python $REPO/scripts/gemm_bench.py --only 0,9 --reps 4 > $OUT/grbm_$D.log 2>&1
done
python - <<PY > $OUT/${TAG}_gemm_sq_counters.txt
import csv, glob, collections
print('SQ counters of dm_gemm_f32 (fp32 MFMA), scripts/gemm_sq_counters.sh: 4096^3 (128x128 tiles) and 2500x1800x1000 (64x64 tiles); ratios to SQ_WAVE_CYCLES')
print('(SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles; GRBM_GUI_ACTIVE / duration = effective clock)')
for d in (1, 0):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f'/tmp/prof_sq_{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'gemm_' not in k or 'splitk' in k: continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    clk = {}
    for f in glob.glob(f'/tmp/prof_grbm_{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'gemm_' not in k or 'splitk' in k: continue
            dur = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
            c = clk.setdefault(k, [0.0, 0.0, 0]); c[0] += float(r['Counter_Value']); c[1] += dur; c[2] += 1
    print(f'--- DM_GEMM_DMA={d}')
    for k, v in agg.items():
        w = v.get('SQ_WAVE_CYCLES', 0) or 1
        n = cnt[(k, 'SQ_WAVE_CYCLES')]
        ghz = clk[k][0] / 8.0 / clk[k][1] if k in clk and clk[k][1] else float('nan')      # GRBM_GUI_ACTIVE sums the 8 XCDs
        # matrix-pipe occupancy: busy cycles of all SIMDs / (1024 SIMDs x the kernel's cycles), both per launch
        cyc = clk[k][0] / 8.0 / clk[k][2] if k in clk and clk[k][2] else float('nan')
        pipe = v['SQ_VALU_MFMA_BUSY_CYCLES'] / n / (1024.0 * cyc) if n and cyc == cyc else float('nan')
        print(f'{k[:96]}: launches {n}  WAIT_ANY {v["SQ_WAIT_ANY"]/w:.3f}  WAIT_INST_ANY {v["SQ_WAIT_INST_ANY"]/w:.3f}  WAIT_INST_LDS {v["SQ_WAIT_INST_LDS"]/w:.3f}  '
              f'ACTIVE_INST_ANY {v["SQ_ACTIVE_INST_ANY"]/w:.3f}  LDS_BANK_CONFLICT {v["SQ_LDS_BANK_CONFLICT"]/w:.4f}  matrix pipe busy {pipe:.3f} of the launch (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / kernel cycles)  '
              f'effective clock {ghz:.2f} GHz (GRBM_GUI_ACTIVE / kernel duration, profiled pass)')
PY
cat $OUT/${TAG}_gemm_sq_counters.txt
