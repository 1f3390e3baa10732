#!/bin/bash
# round 6, first GPU call: the whole -m gpu suite, the parity margins of the full-size tests, a baseline bench line, the CU-reservation
# A/B and the whole-MLP chain threshold at the 300 rows of a 6-column shard.
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/gputest.txt 2>&1; echo "rc $?" >> $OUT/gputest.txt
tail -4 $OUT/gputest.txt
timeout 900 python -m pytest tests -m gpu -s -q -k "atari_literal or dmc_native or autocast or amp_gradients" 2>&1 | grep -v "^$" | grep -iv "warning\|warn(\|autocast(enabled" > $OUT/parity_margins.txt
tail -3 $OUT/parity_margins.txt
line() { python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), 'max', round(d['ms_per_step_max'],3), 'loss', d['loss_model_last'])
"; }
B="--steps 30 --warmup 8 --reps 3 --no-cpu-baseline --no-h2d-leg --prof-steps 0"
{
for cfg in "--dtype f32" "--dtype bf16" "--dtype f32 --pipeline --emulate-world 8"; do
  for v in "X=0" "DM_AC_RESERVE_CUS=4" "DM_AC_RESERVE_CUS=4 DM_WGRAD_SIDE_RESERVE_CUS=4" "DM_AC_RESERVE_CUS=4 DM_WGRAD_SIDE_RESERVE_CUS=4 DM_MAIN_RESERVE_CUS=4" "DM_AC_RESERVE_CUS=8 DM_WGRAD_SIDE_RESERVE_CUS=8" "DM_WGRAD_SIDE_RESERVE_CUS=4"; do
    echo "== $cfg | $v"
    env $v timeout 300 python bench.py $B $cfg 2>/dev/null | line
  done
done
} > $OUT/cu_reserve.txt 2>&1
cat $OUT/cu_reserve.txt
{
for v in 256 1024; do
  for r in 0 7; do
    echo "== 8-way shard of rank $r, DM_CHAIN_MIN_ROWS=$v"
    DM_CHAIN_MIN_ROWS=$v timeout 300 python bench.py $B --pipeline --emulate-world 8 --emulate-rank $r 2>/dev/null | line
  done
done
} > $OUT/chain_rows.txt 2>&1
cat $OUT/chain_rows.txt
python bench.py --no-cpu-baseline --no-h2d-leg > $OUT/bench_base.json 2> $OUT/bench_base.err
python -c "import json; d=json.load(open('$OUT/bench_base.json')); print('bench', d['ms_per_step'], d['ms_per_step_regions'], d['roofline']['frac'], d['roofline']['all_gemm'])"
