#!/bin/bash
OUT=$PWD/gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
for D in ${DBGS:-0 1 2 3}; do
  cd /tmp; rm -rf /tmp/prof_cd_$D
  DM_CHAIN_DBG=$D timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_cd_$D -o t -- python $REPO/bench.py --reps 1 --steps 4 --warmup 3 --no-cpu-baseline --no-h2d-leg --prof-steps 0 --no-overlap > /dev/null 2> $OUT/cd_$D.err
  echo "== DM_CHAIN_DBG=$D (0 normal, 1 loads only, 2 MFMAs only, 3 neither)"
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/prof_cd_$D/**/*kernel_trace.csv',recursive=True)[0]
g=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'mlp_chain_fwd' in r['Kernel_Name']:
        g[next((r[c] for c in ('Grid_Size','Grid_Size_X','grid_size_x') if c in r), str(list(r.keys())))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(g.items()):
    v=sorted(v)[:-2] if len(v)>4 else v
    print('   grid', k, 'calls', len(v), 'median us', round(sorted(v)[len(v)//2],1), 'min', round(min(v),1))
PY
  cd $REPO
done
