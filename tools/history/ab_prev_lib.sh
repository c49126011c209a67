L=stark-perpetual_amd/lib
cp $L/libstarkperp.so /tmp/cur.so
run() { python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-airfri | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['timed_regions']['min_s'], d['timed_regions']['median_s'])"; }
for rep in 1 2 3; do
  cp /tmp/cur.so $L/libstarkperp.so; run current
  cp $L/prev/libstarkperp.so $L/libstarkperp.so; run previous
done
cp /tmp/cur.so $L/libstarkperp.so
