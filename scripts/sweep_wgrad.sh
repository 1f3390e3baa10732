cd $GRAFT_REPO_ROOT
for T in 1 2 3 4 5; do for S in 2 3 4 5 6 7 8 9 10 12 14 16 19 20 24 28 32 38 48; do
echo "== tile $T split $S"; DM_GEMM_TILE=$T DM_GEMM_SPLIT=$S python scripts/gemm_bench.py --only 6,7,19,27 --reps 5 2>/dev/null | grep TF
done; done
