for C in 0 1; do
  for W in 26 27; do
    echo "contiguous=$C w=$W"
    STARKPERP_CONTIGUOUS_TABLES=$C python tools/quick_bulk.py 22 $W 2>&1 | grep -v amdgpu | tail -2
  done
  echo "contiguous=$C lone tree / 20-tree forest (w=26)"
  STARKPERP_CONTIGUOUS_TABLES=$C python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-airfri | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forest', d['value'], d['timed_regions']['min_s'])"
  STARKPERP_CONTIGUOUS_TABLES=$C python bench.py --gpus 1 --steps 1 --warmup 2 --no-cpu-baseline --no-extras --no-airfri | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lone tree ms', d['ms_per_step'], d['timed_regions']['min_s'])"
done
