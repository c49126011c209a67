export TMPDIR=/tmp
O=$PWD/gpurun_out/r03g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pedersen.py tests/test_gpu_state.py -x -q 2>&1 | tail -3 > $O/pytest.txt
for T in 20; do
  rocprofv3 --kernel-trace --output-format csv -d $O/lt$T -o t -- python tools/level_times.py run $T 26 > $O/level_times_$T.txt 2>&1
  python tools/level_times.py parse $O/lt$T/t_kernel_trace.csv 40 > $O/levels_$T.txt 2>&1
done
rm -rf $O/lt20
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-extras --no-airfri --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('20 trees %.3e' % d['value'])"
STARKPERP_FINISH_LANES=131072 python bench.py --no-cpu-baseline --no-extras --no-airfri --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('20 trees, finish lanes 131072 %.3e' % d['value'])"
done
cat $O/pytest.txt; head -10 $O/levels_20.txt
