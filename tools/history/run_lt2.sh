# usage: bash tools/run_lt2.sh <tag> [env assignments...]
export TMPDIR=/tmp
R=$PWD
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p gpurun_out/$TAG
for T in 1 20; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/t$T -o t -- python tools/level_times.py run $T 26 > gpurun_out/$TAG/run_$T.log 2>&1
  f=$(find gpurun_out/$TAG/t$T -name '*kernel_trace.csv' | head -1)
  python tools/level_times.py parse $f 40 > gpurun_out/$TAG/levels_$T.txt 2>&1
  rm -rf gpurun_out/$TAG/t$T
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/$TAG/bench20.json 2> gpurun_out/$TAG/bench20.err
for f in gpurun_out/$TAG/run_1.log gpurun_out/$TAG/run_20.log; do tail -n 2 $f; done
python -c "
import json;d=json.load(open('gpurun_out/$TAG/bench20.json'));print('bench20', d['value'], d['ms_per_step'])"
