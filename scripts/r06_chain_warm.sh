#!/bin/bash
# A/B of the whole-MLP kernel's L2 warm-up (DM_CHAIN_WARM): per-kernel durations from a serial kernel trace + the overlapped step
OUT=$PWD/gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
for W in ${WARMS:-0 1}; do
  cd /tmp; rm -rf /tmp/prof_cw_$W
  DM_CHAIN_WARM=$W timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cw_$W -o t -- python $REPO/bench.py --reps 1 --steps 6 --warmup 3 --no-cpu-baseline --no-h2d-leg --prof-steps 0 --no-overlap ${EXTRA:-} > /dev/null 2> $OUT/cw_$W.err
  echo "== DM_CHAIN_WARM=$W serial kernel stats (name, calls, total ns, avg ns)"
  python - <<PY
import csv,glob
f=glob.glob('/tmp/prof_cw_$W/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('   total kernel time per step (9 steps) ms', round(tot/9/1e6,3))
for r in rows:
    if any(k in r['Name'] for k in ('mlp_chain_fwd','chain_warm','mlp_chain_pack')):
        print('  ', r['Name'][:50], r['Calls'], r['TotalDurationNs'], round(float(r['AverageNs'])/1e3,1),'us')
PY
  cd $REPO
done
for W in ${WARMS:-0 1}; do
  for cfg in "--dtype f32" "--dtype bf16" "--dtype f32 --pipeline --emulate-world 8"; do
    echo "== DM_CHAIN_WARM=$W $cfg"
    DM_CHAIN_WARM=$W timeout 300 python bench.py --steps 30 --warmup 8 --reps 3 --no-cpu-baseline --no-h2d-leg --prof-steps 0 $cfg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), 'max', round(d['ms_per_step_max'],3), 'loss', d['loss_model_last'])
"
  done
done
