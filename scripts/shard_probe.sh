# 7-of-50-column shard (the per-rank workload of an 8-GPU strong-scaling run): eager vs hipGraph replay, plus a kernel timeline
set -u
OUT=$PWD/gpurun_out/shard; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 --emulate-world 8 > $OUT/eager.json 2> $OUT/eager.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 --emulate-world 8 --graph > $OUT/graph.json 2> $OUT/graph.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 --emulate-world 8 --no-overlap > $OUT/noov.json 2> $OUT/noov.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sh -o t -- python $REPO/bench.py --steps 8 --warmup 3 --no-cpu-baseline --prof-steps 0 --emulate-world 8 > $OUT/ks_bench.json 2> $OUT/ks.err
cp $(find /tmp/prof_sh -name "*kernel_stats.csv" | head -1) $OUT/shard_kernel_stats.csv
python $REPO/scripts/trace_timeline.py $(find /tmp/prof_sh -name "*kernel_trace.csv" | head -1) 2 > $OUT/shard_timeline.txt 2>&1
cd $REPO
python - << PY
import json
for n in ('eager','graph','noov'):
    try:
        d=json.load(open('$OUT/%s.json'%n)); print(n, d['ms_per_step'], d.get('host_enqueue_ms_per_step'))
    except Exception as e: print(n, 'failed', e)
PY
head -30 $OUT/shard_timeline.txt
