# round 4, third GPU pass: r04 profile set, per-level timelines of 20- / 10- / 5-tree forests (the tail that a
# two-group co-schedule could hide), the whole GPU suite
export TMPDIR=/tmp
O=gpurun_out/r04c; mkdir -p $O
bash tools/run_r04_prof.sh > $O/prof.log 2>&1
for T in 20 10 5 1; do
  rocprofv3 --kernel-trace --output-format csv -d $O/lt$T -o t -- python tools/level_times.py run $T 26 > $O/lt$T.txt 2>&1
  N=$(python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$O/lt$T/t_kernel_trace.csv")) if "ped_" in r["Kernel_Name"]]
print(len(rows)//8)
PY
)
  python tools/level_times.py parse $O/lt$T/t_kernel_trace.csv $N > $O/levels_forest_$T.txt 2>&1
  rm -rf $O/lt$T
done
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
tail -3 $O/levels_forest_20.txt $O/levels_forest_10.txt
