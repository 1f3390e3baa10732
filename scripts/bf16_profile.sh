set -u
OUT=$PWD/gpurun_out/bf16prof; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o t -- python $REPO/bench.py --dtype bf16 --steps 8 --warmup 3 --no-cpu-baseline --prof-steps 0 > $OUT/ks_bench.json 2> $OUT/ks.err
cp $(find /tmp/prof_ks -name "*kernel_stats.csv" | head -1) $OUT/bf16_kernel_stats.csv
python $REPO/scripts/trace_timeline.py $(find /tmp/prof_ks -name "*kernel_trace.csv" | head -1) 2 > $OUT/bf16_timeline.txt 2>&1
head -45 $OUT/bf16_timeline.txt
