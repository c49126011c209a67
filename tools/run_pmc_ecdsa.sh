export STARKPERP_WINDOW_BITS=26
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_IFETCH"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_k/$n -o k -- python $R/tools/quick_ecdsa.py 16 2 > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_k/*/k_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "ecdsa_verify" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k)
    for c,vals in sorted(v.items()): print("   %-24s n=%d last=%.4g" % (c,len(vals),vals[-1]))
PY
