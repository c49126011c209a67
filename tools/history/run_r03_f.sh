export TMPDIR=/tmp
O=$PWD/gpurun_out/r03f
mkdir -p $O
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
  rm -rf $O/pc
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pc -o t -- python tools/quick_lde.py > /dev/null 2>&1
  python - <<PY
import csv, collections
acc = collections.OrderedDict()
try:
    for r in csv.DictReader(open("$O/pc/t_counter_collection.csv")):
        if "ntt_tile" not in r["Kernel_Name"]: continue
        d = acc.setdefault(int(r["Dispatch_Id"]), {"grid": r["Grid_Size"]})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    for k, d in list(acc.items())[-5:]:
        print(" ".join("%s=%s" % (a, ("%.3e" % b) if isinstance(b, float) else b) for a, b in d.items()))
except Exception as e:
    print("failed:", "$SET", e)
PY
done
rm -rf $O/pc
