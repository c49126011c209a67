# round 3, GPU run B: parity of the hash / tree suites + per-level traces + the headline number
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r03b}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pedersen.py tests/test_gpu_state.py -m gpu -x -q > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
for T in 1 20; do
  rocprofv3 --kernel-trace --output-format csv -d $O/lt$T -o t -- python tools/level_times.py run $T 26 > $O/level_times_$T.txt 2>&1
  N=$([ $T = 1 ] && echo 17 || echo 24)
  python tools/level_times.py parse $O/lt$T/t_kernel_trace.csv $N > $O/levels_$T.txt 2>&1
  grep forest $O/level_times_$T.txt
done
rm -rf $O/lt1 $O/lt20
python bench.py --no-cpu-baseline --no-extras --no-airfri > $O/bench_quick.json 2> $O/bench_quick.err
python -c "
import json; d=json.load(open('$O/bench_quick.json')); print(d['value'], d['ms_per_step'], d['timed_regions'], d['roofline']['frac'], d['roofline']['whole_region']['frac'])"
