# rocprofv3 passes behind profiles/r04_*: kernel stats of the driver's command, PMC traffic of the same command
# (separate FETCH_SIZE / WRITE_SIZE passes), SQ_INSTS_VALU of the bulk kernels, the airfri workload.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04prof
mkdir -p $O
CMD="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-airfri"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- $CMD > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o b -- $CMD > $O/fetch_bench.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o b -- $CMD > $O/write_bench.json 2> $O/write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -o b -- python tools/bulk_only.py 22 26 > $O/sq.log 2>&1
python tools/valu_counts.py $O/sq/b_counter_collection.csv $O/sq/b_kernel_trace.csv 22 26 > $O/valu_counts_w26.json
NL=$(python -c "import json;print(json.load(open('$O/stats_bench.json'))['roofline']['launches'])")
python tools/trace_timed_avg.py $O/stats/b_kernel_trace.csv $NL > $O/timed_region_launch_avg.txt 2>&1
NLF=$(python -c "import json;print(json.load(open('$O/fetch_bench.json'))['roofline']['launches'])")
KEY=$(python -c "import json;d=json.load(open('$O/fetch_bench.json'));c=d['config'];print('merkle:steps=%d:calls=%s:streams=%d:w=%d'%(d['steps'],','.join(map(str,c['timed_calls'])),c['streams'],c['window_bits']))")
PMC_TIMED_LAUNCHES=$NLF python tools/pmc_traffic.py $O/fetch/b_counter_collection.csv $O/write/b_counter_collection.csv r04_pmc_traffic.json "$KEY" "bench.py --gpus 1 --steps 20 --warmup 5 (round 4: regions repeated until 50 ms; the pure ped_accumulate_kernel launches are levels 0 and 1 of every 20-tree forest), 26-bit windows" > /dev/null
cp profiles/r04_pmc_traffic.json $O/
# airfri workload
rocprofv3 --kernel-trace --stats --output-format csv -d $O/airfri -o a -- python bench.py --workload airfri --steps 6 --warmup 3 --no-cpu-baseline > $O/airfri_bench.json 2> $O/airfri.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/afetch -o a -- python bench.py --workload airfri --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/awrite -o a -- python bench.py --workload airfri --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_traffic.py $O/afetch/a_counter_collection.csv $O/awrite/a_counter_collection.csv r04_pmc_traffic_airfri.json airfri "bench.py --workload airfri --steps 3 --warmup 1" > /dev/null
cp profiles/r04_pmc_traffic_airfri.json $O/
cp $O/stats/b_kernel_stats.csv $O/r04_kernel_stats.csv
cp $O/airfri/a_kernel_stats.csv $O/r04_airfri_kernel_stats.csv
grep -h "ped_\|sp::" $O/fetch/b_counter_collection.csv | head -400 > $O/bench_steps20_fetch_hash_kernels.csv
grep -h "ped_\|sp::" $O/write/b_counter_collection.csv | head -400 > $O/bench_steps20_write_hash_kernels.csv
grep -h "ntt_tile\|air_eval\|fri_fold" $O/afetch/a_counter_collection.csv | head -300 > $O/airfri_fetch_prover_kernels.csv
grep -h "ntt_tile\|air_eval\|fri_fold" $O/awrite/a_counter_collection.csv | head -300 > $O/airfri_write_prover_kernels.csv
head -1 $O/sq/b_counter_collection.csv > $O/bulk_2p22_sq_counters_w26.csv; grep "ped_" $O/sq/b_counter_collection.csv >> $O/bulk_2p22_sq_counters_w26.csv
rm -rf $O/stats $O/fetch $O/write $O/sq $O/airfri $O/afetch $O/awrite
ls -la $O; cat $O/timed_region_launch_avg.txt $O/valu_counts_w26.json
