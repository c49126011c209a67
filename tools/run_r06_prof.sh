#!/bin/bash
# The ONE scripted evidence pass of round 6.  Run on the GPU box AFTER the last code commit:
#     bash tools/run_r06_prof.sh        -> gpurun_out/r06prof/*, copied into profiles/ by the builder
# Every figure bench.py prints from a file under profiles/ (instr_per_hash, traffic) and every profile the judge reads
# (kernel stats, level counters, C3 counters / timeline) comes from this pass, and r06_evidence.json carries the hashes
# of the library and of the kernel sources it ran on (tests/test_evidence_fresh_cpu.py compares them with the tree).
set -u
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"
O=$R/gpurun_out/r06prof
rm -rf $O; mkdir -p $O
cd $R
python tools/evidence_stamp.py > $O/r06_evidence.json
T0=$SECONDS
note() { echo "[r06prof +$((SECONDS - T0))s] $*" | tee -a $O/log.txt; }

# ---- 1. kernel trace + stats of the driver's own command --------------------------------------------------------
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 \
    > $O/r06_stats_bench.json 2> $O/stats.err
cd $R
cp bench_detail.json $O/r06_stats_bench_detail.json
NL=$(python -c "import json;print(json.load(open('$O/r06_stats_bench.json'))['roofline']['launches'])")
KT=$(find $O/stats -name "*kernel_trace.csv" | head -1); KS=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python tools/trace_timed_avg.py $KT $NL > $O/r06_timed_region_launch_avg.txt 2>&1
cp $KS $O/r06_kernel_stats.csv
note "kernel stats done ($NL timed launches)"

# ---- 2. the same command without a profiler: the line and the detail the judge compares with the driver's -------
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_default.json 2> $O/bench_default.err
cp bench_detail.json $O/r06_bench_detail.json
note "plain bench done: $(wc -c < $O/r06_bench_default.json) bytes"

# ---- 3. PMC traffic, merkle workload (separate FETCH_SIZE / WRITE_SIZE passes; shorter windows: counters serialise) --
CMD="python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-airfri --min-timed-s 0.25 --preheat-s 0.25"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o b -- $CMD > $O/fetch_bench.json 2> $O/fetch.err
cp $R/bench_detail.json $O/fetch_detail.json
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o b -- $CMD > $O/write_bench.json 2> $O/write.err
cd $R
NLF=$(python -c "import json;print(json.load(open('$O/fetch_bench.json'))['roofline']['launches'])")
KEY=$(python -c "import json;d=json.load(open('$O/fetch_detail.json'));c=d['config'];print('merkle:steps=%d:calls=%s:streams=%d:w=%d'%(d['steps'],','.join(map(str,c['timed_calls'])),c['streams'],c['window_bits']))")
FC=$(find $O/fetch -name "*counter_collection.csv" | head -1); WC=$(find $O/write -name "*counter_collection.csv" | head -1)
PMC_TIMED_LAUNCHES=$NLF python tools/pmc_traffic.py $FC $WC r06_pmc_traffic.json "$KEY" "bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-airfri --min-timed-s 0.25 --preheat-s 0.25 (round 6 evidence pass: the sustained line with shorter windows - PMC passes serialise the dispatches; the pure ped_accumulate_kernel launches are levels 0 and 1 of every 20-tree forest), 26-bit windows" > /dev/null
cp profiles/r06_pmc_traffic.json $O/
note "merkle PMC traffic done (key $KEY)"

# ---- 4. PMC traffic, airfri workload -------------------------------------------------------------------------------
CMDA="python $R/bench.py --workload airfri --steps 3 --warmup 1 --no-cpu-baseline"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/afetch -o a -- $CMDA > $O/afetch_bench.json 2> $O/afetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/awrite -o a -- $CMDA > $O/awrite_bench.json 2> $O/awrite.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/astats -o a -- python $R/bench.py --workload airfri --steps 6 --warmup 3 --no-cpu-baseline > $O/r06_airfri_stats_bench.json 2> $O/astats.err
cd $R
AFC=$(find $O/afetch -name "*counter_collection.csv" | head -1); AWC=$(find $O/awrite -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $AFC $AWC r06_pmc_traffic_airfri.json "airfri:rows=2^20:streams=3:w=26" "bench.py --workload airfri --steps 3 --warmup 1 (round 6 evidence pass), 26-bit windows" > /dev/null
cp profiles/r06_pmc_traffic_airfri.json $O/
cp $(find $O/astats -name "*kernel_stats.csv" | head -1) $O/r06_airfri_kernel_stats.csv
note "airfri PMC traffic + stats done"

# ---- 5. SQ_INSTS_VALU per hash of the bulk kernels at both window plans ------------------------------------------
cd /tmp
for w in 26 21; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq$w -o b -- python $R/tools/bulk_only.py 22 $w > /dev/null 2> $O/sq$w.err
  python $R/tools/valu_counts.py $(find $O/sq$w -name "*counter_collection.csv" | head -1) $(find $O/sq$w -name "*kernel_trace.csv" | head -1) 22 $w > $O/valu_counts_w$w.json 2>> $O/sq$w.err
done
# ---- 6. level counters and level times of the 20-tree forest -------------------------------------------------------
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/lc -o t -- python $R/tools/level_times.py run 20 26 > $O/lc_run.txt 2> $O/lc.err
cd $R
python tools/level_counters.py $(find $O/lc -name "*counter_collection.csv" | head -1) 15 > $O/r06_level_counters_forest_20.txt 2>> $O/lc.err
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/lt -o t -- env LEVEL_TIMES_SUSTAINED=1 python $R/tools/level_times.py run 20 26 > $O/lt_run.txt 2> $O/lt.err
cd $R
python tools/level_times.py parse $(find $O/lt -name "*kernel_trace.csv" | head -1) 15 > $O/r06_levels_forest_20.txt 2>> $O/lt.err
cat $O/lt_run.txt >> $O/r06_levels_forest_20.txt
python tools/make_valu_issue.py $O/valu_counts_w26.json $O/valu_counts_w21.json $O/r06_level_counters_forest_20.txt $O/r06_evidence.json > $O/r06_valu_issue.json 2> $O/valu_issue.err
note "instruction counts + level counters done"

# ---- 7. configs[2]: calls, timeline, counters ------------------------------------------------------------------------
bash tools/run_r06_c3.sh > $O/c3.log 2>&1
cp gpurun_out/r06c3/calls.txt $O/r06_c3_calls.txt
cp gpurun_out/r06c3/host_timeline.txt $O/r06_c3_host_timeline.txt
cp gpurun_out/r06c3/r06_c3_timeline.txt gpurun_out/r06c3/r06_c3_sq_counters.json $O/
note "C3 done"

# ---- 8. with the new instruction counts / traffic files in place: the line once more (what bench.py prints from them) --
cp $O/r06_valu_issue.json $O/r06_pmc_traffic.json $O/r06_pmc_traffic_airfri.json $O/r06_c3_sq_counters.json profiles/
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_default.json 2> $O/bench_default.err
cp bench_detail.json $O/r06_bench_detail.json
note "final line: $(wc -c < $O/r06_bench_default.json) bytes"
rm -rf $O/stats $O/fetch $O/write $O/afetch $O/awrite $O/astats $O/sq26 $O/sq21 $O/lc $O/lt
ls -la $O; cat $O/r06_timed_region_launch_avg.txt; head -12 $O/r06_kernel_stats.csv | cut -c1-160
