#!/bin/bash
# Round profile set, run on the GPU box from the repo root:  bash scripts/collect_profiles.sh r03
#   1. rocprofv3 --kernel-trace --stats of the default bench command's workload (short run) -> <tag>_kernel_stats.csv
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass; counters only, no tracing domains besides
#      --kernel-trace) of a 3-step single-stream run -> <tag>_pmc_traffic.json (carries the kernel-source fingerprint)
#   3. the bench line itself, reading the fresh PMC file -> <tag>_bench.json
# Everything lands in gpurun_out/<tag>/ (scratch); copy what should be judged into profiles/.
# PARTS (default "core configs lab") selects sections: core = kernel statistics + counter passes + the bench line; configs = the
# other configurations (shards, atari-native, dmc, bf16 with its own counter passes), per-shape table, shard statistics, queue
# timelines; lab = the tile-kernel laboratory, SQ counters, tile sweep, persistent-kernel phase clocks (code-specific: re-run when
# csrc/gemm.hip or csrc/rssm_lds.hip changed).  Every profiler run is bounded by `timeout`.
set -u
TAG=${1:-r06}
PARTS=${PARTS:-core configs lab}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
RP="timeout 600 rocprofv3"
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
SHA=$(python bench.py --csrc-sha)
cd /tmp
if has core; then
$RP --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o t -- python $REPO/bench.py --reps 1 --steps 8 --warmup 3 --no-cpu-baseline --no-h2d-leg --prof-steps 0 > $OUT/ks_bench.json 2> $OUT/ks.err
cp $(find /tmp/prof_ks -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
python $REPO/scripts/trace_timeline.py $(find /tmp/prof_ks -name "*kernel_trace.csv" | head -1) 2 > $OUT/${TAG}_timeline.txt 2>&1
# the same workload on ONE stream (--no-overlap): per-kernel durations undisturbed by concurrent streams - the file the bench line's
# roofline.achieved (HIP events in a single-stream pass) must agree with
$RP --kernel-trace --stats --output-format csv -d /tmp/prof_ks_serial -o t -- python $REPO/bench.py --reps 1 --steps 8 --warmup 3 --no-cpu-baseline --no-h2d-leg --prof-steps 0 --no-overlap > $OUT/ks_serial_bench.json 2> $OUT/ks_serial.err
cp $(find /tmp/prof_ks_serial -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_serial.csv
for C in FETCH_SIZE WRITE_SIZE; do
  $RP --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -o p -- python $REPO/bench.py --reps 1 --steps 2 --warmup 1 --no-overlap --no-cpu-baseline --no-h2d-leg --prof-steps 0 > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
python $REPO/scripts/pmc_traffic.py $(find /tmp/prof_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/prof_WRITE_SIZE -name "*counter_collection.csv" | head -1) 3 $SHA > $OUT/${TAG}_pmc_traffic.json
cd $REPO
cp $OUT/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json      # where bench.py looks by default (on this box; copy gpurun_out/<tag>/ into profiles/ afterwards)
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
fi
cd $REPO
if has configs; then
python bench.py --reps 1 --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 --pipeline --emulate-world 8 > $OUT/${TAG}_bench_shard7of50.json 2>/dev/null
for W in 2 4; do python bench.py --reps 1 --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 --pipeline --emulate-world $W > $OUT/${TAG}_bench_shard_world$W.json 2>/dev/null; done
python bench.py --reps 1 --steps 20 --warmup 5 --no-cpu-baseline --workload atari-native > $OUT/${TAG}_bench_atari_native.json 2>/dev/null
python bench.py --reps 1 --steps 10 --warmup 4 --no-cpu-baseline --workload dmc --dtype bf16 > $OUT/${TAG}_bench_dmc_bf16.json 2>/dev/null
python bench.py --reps 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d-leg --prof-steps 0 --pipeline --dtype bf16 --emulate-world 8 > $OUT/${TAG}_bench_shard7of50_bf16.json 2>/dev/null
bash scripts/collect_pmc_bf16.sh $TAG > $OUT/collect_bf16.log 2>&1      # bf16 step: its own counter passes, then the bf16 bench line
DM_BF16_NO_TWINS=1 python bench.py --dtype bf16 --no-cpu-baseline --no-h2d-leg --pmc-json /nonexistent > $OUT/${TAG}_bench_bf16_fp32_storage.json 2>/dev/null
python bench.py --reps 1 --steps 10 --warmup 4 --no-cpu-baseline --no-h2d-leg --pmc-json /nonexistent --shape-table $OUT/${TAG}_gemm_shapes.txt > /dev/null 2>&1
# kernel statistics of the 7-column shard (the rollout / posterior / BPTT chains at 350 rows: DESIGN 6)
cd /tmp; $RP --kernel-trace --stats --output-format csv -d /tmp/prof_ks_shard -o t -- python $REPO/bench.py --reps 1 --steps 8 --warmup 3 --no-cpu-baseline --no-h2d-leg --prof-steps 0 --emulate-world 8 --no-overlap > /dev/null 2> $OUT/ks_shard.err
cp $(find /tmp/prof_ks_shard -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_shard7of50_serial.csv; cd $REPO
# per-queue timelines of one step (which stream is busy when), fp32 and bf16
bash scripts/gpu_trace.sh f32 $TAG > /dev/null 2>&1; cp $OUT/queues_f32.txt $OUT/${TAG}_queues_f32.txt
bash scripts/gpu_trace.sh bf16 $TAG > /dev/null 2>&1; cp $OUT/queues_bf16.txt $OUT/${TAG}_queues_bf16.txt
fi
if has lab; then
# the tile-kernel laboratory (LDS-DMA loop variants, ablations, in-kernel clock probe; production dm_gemm_f32 beside them), the SQ
# counters, the tile sweep, the persistent posterior kernel's phase clocks
LAB_REPS=20 bash scripts/microbench/run_gemm_lab.sh > /dev/null 2>&1; cp gpurun_out/gemm_lab.txt $OUT/${TAG}_gemm_lab.txt
bash scripts/gemm_sq_counters.sh $TAG > /dev/null 2>&1
python scripts/gemm_tile_sweep.py > $OUT/${TAG}_tile_sweep.txt 2>&1
python scripts/persist_prof.py 50 25 13 7 2>/dev/null | grep -v Warning > $OUT/${TAG}_rssm_lds.txt
fi
# (the per-CU load-rate microbenchmark, scripts/microbench/l2_stream.hip, concerns code that has not changed since it was taken:
#  profiles/<tag>_l2_stream.txt is kept from the earlier collection)
python - << PY
import json, os
if not os.path.exists('$OUT/${TAG}_bench_bf16.json'):
    raise SystemExit(0)
d = json.load(open('$OUT/${TAG}_bench.json'))
print('ms/step', d['ms_per_step'], 'value', d['value'], 'roofline', {k: d['roofline'][k] for k in ('kernel', 'achieved', 'frac', 'traffic', 'launches_per_step')})
print('cpu', d['cpu_baseline'])
p = json.load(open('$OUT/${TAG}_pmc_traffic.json'))
print('HBM GB/step', p['total_gb_per_step'])
s = json.load(open('$OUT/${TAG}_bench_shard7of50.json'))
print('shard 7 of 50:', s['ms_per_step'])
b = json.load(open('$OUT/${TAG}_bench_bf16.json'))
print('bf16:', b['ms_per_step'], b['value'])
PY
head -3 $OUT/${TAG}_timeline.txt
