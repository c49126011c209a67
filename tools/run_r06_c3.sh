#!/bin/bash
# configs[2] as a measured pipeline.  On the GPU box: bash tools/run_r06_c3.sh
#   1. 24 sp_order_batch calls, every call's time (median / p90);
#   2. rocprofv3 kernel trace of one call -> timeline (chains, verification, tree levels, overlap);
#   3. SQ_INSTS_VALU / SQ_WAVES of the four kernels -> instructions per signature / per wave.
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"
O=$R/gpurun_out/r06c3
mkdir -p $O
export TMPDIR=/tmp
cd $R
python tools/c3_probe.py calls 24 > $O/calls.txt 2> $O/calls.err
python tools/c3_probe.py calls 24 >> $O/calls.txt 2>> $O/calls.err
STARKPERP_TIMELINE=1 python tools/c3_probe.py calls 8 > /dev/null 2> $O/host_timeline.txt
sed -i "/amdgpu.ids/d" $O/host_timeline.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/tools/c3_probe.py one > $O/one.txt 2> $O/one.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc -o p -- python $R/tools/c3_probe.py pmc > $O/pmc.txt 2> $O/pmc.err
cd $R
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/c3_timeline.py $T > $O/r06_c3_timeline.txt 2>&1
C=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python tools/c3_counters.py $C $O/r06_c3_sq_counters.json > /dev/null 2> $O/counters.err
rm -rf $O/trace $O/pmc
cat $O/calls.txt; tail -25 $O/r06_c3_timeline.txt; head -60 $O/r06_c3_sq_counters.json
