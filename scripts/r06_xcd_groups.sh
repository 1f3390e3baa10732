#!/bin/bash
# 2-D XCD grouping of the tile order (DM_GEMM_XCD_GROUPS: 1 = off, 0 = auto): time and FETCH_SIZE of the rollout / head products
OUT=$PWD/gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
for G in 1 0; do
  echo "== DM_GEMM_XCD_GROUPS=$G  (TF/s lines of scripts/gemm_bench.py)"
  DM_GEMM_XCD_GROUPS=$G python scripts/gemm_bench.py --only 0,3,9,10,8,22 --reps 20 2>/dev/null | grep -i "TF"
  cd /tmp; rm -rf /tmp/prof_xg_$G
  DM_GEMM_XCD_GROUPS=$G rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_xg_$G -o p -- python $REPO/scripts/gemm_bench.py --only 0,3,9,10,8,22 --reps 4 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/prof_xg_$G/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'gemm' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
        agg[(r['Kernel_Name'][:60], r['Grid_Size'])].append(float(r['Counter_Value']))
for k,v in agg.items():
    print('   FETCH_SIZE x2 MB per launch', round(2*1024*sum(v)/len(v)/1e6,1), 'launches', len(v), k)
PY
  cd $REPO
done
bash scripts/r06_quick.sh xcd_auto
DM_GEMM_XCD_GROUPS=1 bash scripts/r06_quick.sh xcd_off
