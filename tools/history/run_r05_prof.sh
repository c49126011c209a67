# rocprofv3 passes behind profiles/r05_*: kernel stats of the driver's command (the SUSTAINED line of round 5), PMC
# traffic of the same workload (separate FETCH_SIZE / WRITE_SIZE passes, shorter windows: counters serialise the
# dispatches), the airfri workload.  SQ_INSTS_VALU per hash was not re-collected: csrc/pedersen.hip is unchanged
# since profiles/r04_valu_issue.json.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05prof
mkdir -p $O
CMD="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-airfri"
SHORT="--min-timed-s 0.25 --preheat-s 0.25"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- $CMD > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o b -- $CMD $SHORT > $O/fetch_bench.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o b -- $CMD $SHORT > $O/write_bench.json 2> $O/write.err
NL=$(python -c "import json;print(json.load(open('$O/stats_bench.json'))['roofline']['launches'])")
python tools/trace_timed_avg.py $O/stats/b_kernel_trace.csv $NL > $O/timed_region_launch_avg.txt 2>&1
NLF=$(python -c "import json;print(json.load(open('$O/fetch_bench.json'))['roofline']['launches'])")
KEY=$(python -c "import json;d=json.load(open('$O/fetch_bench.json'));c=d['config'];print('merkle:steps=%d:calls=%s:streams=%d:w=%d'%(d['steps'],','.join(map(str,c['timed_calls'])),c['streams'],c['window_bits']))")
PMC_TIMED_LAUNCHES=$NLF python tools/pmc_traffic.py $O/fetch/b_counter_collection.csv $O/write/b_counter_collection.csv r05_pmc_traffic.json "$KEY" "bench.py --gpus 1 --steps 20 --warmup 5 --min-timed-s 0.25 --preheat-s 0.25 (round 5: the sustained line with shorter windows - PMC passes serialise the dispatches; the pure ped_accumulate_kernel launches are levels 0 and 1 of every 20-tree forest), 26-bit windows" > /dev/null
cp profiles/r05_pmc_traffic.json $O/
rocprofv3 --kernel-trace --stats --output-format csv -d $O/airfri -o a -- python bench.py --workload airfri --steps 6 --warmup 3 --no-cpu-baseline > $O/airfri_bench.json 2> $O/airfri.err
cp $O/stats/b_kernel_stats.csv $O/r05_kernel_stats.csv
cp $O/airfri/a_kernel_stats.csv $O/r05_airfri_kernel_stats.csv
rm -rf $O/stats $O/fetch $O/write $O/airfri
ls -la $O; cat $O/timed_region_launch_avg.txt; head -5 $O/r05_kernel_stats.csv | cut -c1-200
