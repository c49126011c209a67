export TMPDIR=/tmp
O=gpurun_out/r04i; mkdir -p $O
(python tools/quick_init.py 26; STARKPERP_TABLE_BUILD=direct python tools/quick_init.py 26) 2>/dev/null > $O/table_build.txt; cat $O/table_build.txt
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st -o t -- python $GRAFT_REPO_ROOT/tools/quick_init.py 26 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
head -8 $O/st/*/t_kernel_stats.csv 2>/dev/null | cut -c1-200 || find $O/st -name "*stats*" | head
python - <<PY
import csv,glob
f=glob.glob("$O/st/**/t_kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "extend_window" in r["Kernel_Name"]]
import collections
by=collections.defaultdict(list)
for r in rows: by[int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for g in sorted(by): print(g, len(by[g]), "launches, avg %.1f us"%(sum(by[g])/len(by[g])))
PY
rm -rf $O/st
