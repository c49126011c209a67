export TMPDIR=/tmp
O=$PWD/gpurun_out/r03e
mkdir -p $O
python tools/quick_lde.py > $O/lde.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o t -- python tools/quick_lde.py > /dev/null 2>&1
python - <<PY
import csv
rows = [r for r in csv.DictReader(open("$O/kt/t_kernel_trace.csv")) if "ntt_tile" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-5:]:
    print("grid %9s  %8.1f us" % (r["Grid_Size"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pc -o t -- python tools/quick_lde.py > /dev/null 2>&1
python - <<PY
import csv, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open("$O/pc/t_counter_collection.csv")):
    if "ntt_tile" not in r["Kernel_Name"]: continue
    d = acc.setdefault(int(r["Dispatch_Id"]), {"grid": r["Grid_Size"]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
for k, d in list(acc.items())[-5:]:
    w = d.get("SQ_WAVES", 1) or 1
    print("grid %9s waves %7d valu/wave %8.0f lds/wave %7.0f salu/wave %7.0f" % (d["grid"], w, d.get("SQ_INSTS_VALU", 0) / w, d.get("SQ_INSTS_LDS", 0) / w, d.get("SQ_INSTS_SALU", 0) / w))
PY
cat $O/lde.txt
rm -rf $O/kt $O/pc
