# same-box comparison of library builds placed in stark-perpetual_amd/lib/alt/*.so (bulk rate at 26- / 27-bit windows,
# the driver's forest)
L=stark-perpetual_amd/lib
cp $L/libstarkperp.so /tmp/cur.so
for rep in 1 2; do
for V in $L/alt/*.so; do
  cp $V $L/libstarkperp.so
  echo "== $(basename $V) rep $rep"
  python tools/quick_bulk.py 22 26 2>&1 | grep window_bits
  python tools/quick_bulk.py 22 27 2>&1 | grep window_bits
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-airfri | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forest', d['value'], d['timed_regions']['min_s'])"
done
done
cp /tmp/cur.so $L/libstarkperp.so
