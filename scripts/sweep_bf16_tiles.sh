# GPU microbenchmark: bf16-operand tiled GEMM on the step's shapes with each tile forced (DM_GEMM_TILE=1..5) vs the cost model's pick
cd $GRAFT_REPO_ROOT
echo "== model"; python scripts/gemm_bench.py --bf16 --reps 20 2>/dev/null | grep TF
for T in 1 2 3 4 5; do echo "== tile $T"; DM_GEMM_TILE=$T python scripts/gemm_bench.py --bf16 --reps 20 2>/dev/null | grep TF; done
