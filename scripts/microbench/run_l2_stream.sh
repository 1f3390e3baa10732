cd $GRAFT_REPO_ROOT/scripts/microbench
B=./l2_stream
for kib in 512 2048 8192 65536; do
 for cfg in "8 256 1" "8 256 4" "16 256 4" "8 1024 1" "8 1024 2" "32 256 2"; do
  $B $kib $cfg 0
 done
done
echo "--- pattern 1 (fragment gather)"
for kib in 2048 8192; do for cfg in "8 256 1" "8 256 4" "16 256 2" "8 1024 1"; do $B $kib $cfg 1; done; done
echo "--- one CU only"
for kib in 512 2048 8192; do $B $kib 8 256 1 0 1; $B $kib 16 256 4 0 1; $B $kib 8 1024 1 1 1; done
