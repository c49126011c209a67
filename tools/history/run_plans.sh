export TMPDIR=/tmp
mkdir -p gpurun_out/plans
for cfg in "20 1" "10,10 2" "5,5,5,5 4" "5,5,5,5 2" "7,7,6 3" "4,4,4,4,4 5" "2,2,2,2,2,2,2,2,2,2 5" ; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --plan $1 --streams $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', '$2', '%.3e'%d['value'], d['ms_per_step'])" >> gpurun_out/plans/out.txt
done
cat gpurun_out/plans/out.txt
