# where does the bulk kernel wait?  PMC passes of quick_bulk with random inputs (gathers from HBM) and with 2^10 distinct
# pairs (gathers from L2): the counters that differ name the limiter
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04k; mkdir -p $O
R=$PWD
cd /tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_WAVES" "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TA_TCP_STATE_READ TCP_GATE_EN1 TCP_GATE_EN2 TA_BUSY_avr TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr"; do
  tag=$(echo $set | cut -d' ' -f1)
  for mode in rand few10; do
    arg=""; [ $mode = few10 ] && arg="few10"
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag.$mode -o p -- python $R/tools/quick_bulk.py 22 26 $arg > $O/$tag.$mode.log 2>&1
  done
done
cd $R
python - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$O/*.rand")+glob.glob("$O/*.few10")):
    f=glob.glob(d+"/**/p_counter_collection.csv",recursive=True)
    if not f: print(d,"no counters"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "ped_accumulate_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d.split("/")[-1], {k: "%.4g"%(sum(v[-3:])/len(v[-3:])) for k,v in acc.items()})
PY
rm -rf $O/*.rand $O/*.few10
