# A/B of the one-hot z_mlp gather-sum (dm_z_embed_launch) against the z_mlp product: DM_RSSM_NO_Z_EMBED=1 is the product
for e in 0 1; do
  if [ $e = 1 ]; then export DM_RSSM_NO_Z_EMBED=1; fi
  python bench.py --steps 15 --warmup 4 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_embed=$e fp32', d['ms_per_step'])"
  python bench.py --dtype bf16 --steps 15 --warmup 4 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_embed=$e bf16', d['ms_per_step'])"
  python bench.py --steps 15 --warmup 4 --no-cpu-baseline --prof-steps 0 --emulate-world 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_embed=$e shard', d['ms_per_step'])"
done
