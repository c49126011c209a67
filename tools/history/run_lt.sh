set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/lt
for T in 1 20; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lt/t$T -o t -- python tools/level_times.py run $T 26 > gpurun_out/lt/run_$T.log 2>&1
  f=$(find gpurun_out/lt/t$T -name '*kernel_trace.csv' | head -1)
  python tools/level_times.py parse $f 40 > gpurun_out/lt/levels_$T.txt 2>&1
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/lt/bench20.json 2> gpurun_out/lt/bench20.err
tail -3 gpurun_out/lt/run_*.log
