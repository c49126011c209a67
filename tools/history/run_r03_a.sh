# round 3, GPU run A: baseline per-level traces + SQ counters (lone tree, forest of 20) and the
# inversion micro-benchmarks (divsteps on the scalar unit)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03a
mkdir -p $O
./tools/ubench/lat_parts > $O/lat_parts.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/lat -o l -- ./tools/ubench/lat_parts > /dev/null 2>&1
python - > $O/lat_parts_counters.txt <<PY
import csv, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open("$O/lat/l_counter_collection.csv")):
    d = acc.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": r["Grid_Size"], "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
for k, d in acc.items():
    w = d.get("SQ_WAVES", 1) or 1
    print("%-40s grid %7s  %8.1f us  valu/wave/rep %8.0f  salu/wave/rep %8.0f" % (d["name"][:40], d["grid"], d["t"], d.get("SQ_INSTS_VALU", 0) / w / 8, d.get("SQ_INSTS_SALU", 0) / w / 8))
PY
for T in 1 20; do
  rocprofv3 --kernel-trace --output-format csv -d $O/lt$T -o t -- python tools/level_times.py run $T 26 > $O/level_times_$T.txt 2>&1
  N=$([ $T = 1 ] && echo 16 || echo 20)
  python tools/level_times.py parse $O/lt$T/t_kernel_trace.csv $((2 * N)) > $O/levels_$T.txt 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/lc$T -o t -- python tools/level_times.py run $T 26 > /dev/null 2>&1
  python tools/level_counters.py $O/lc$T/t_counter_collection.csv $N > $O/level_counters_$T.txt 2>&1
done
rm -rf $O/lat $O/lt1 $O/lt20 $O/lc1 $O/lc20
ls -la $O
