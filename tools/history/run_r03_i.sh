# end of round 3, after ped_path_kernel / ped_top_kernel: kernel-trace stats of the driver's command and of the airfri
# workload re-collected (the PMC passes of run_r03_prof.sh concern ped_accumulate_kernel, which did not change)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03prof_i
mkdir -p $O
CMD="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-airfri"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- $CMD > $O/stats_bench.json 2> $O/stats.err
NL=$(python -c "import json;print(json.load(open('$O/stats_bench.json'))['roofline']['launches'])")
python tools/trace_timed_avg.py $O/stats/b_kernel_trace.csv $NL > $O/timed_region_launch_avg.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/airfri -o a -- python bench.py --workload airfri --steps 6 --warmup 3 --no-cpu-baseline > $O/airfri_bench.json 2> $O/airfri.err
cp $O/stats/b_kernel_stats.csv $O/r03_kernel_stats.csv
cp $O/airfri/a_kernel_stats.csv $O/r03_airfri_kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $O/lone -o l -- python tools/level_times.py run 1 26 > /dev/null 2>&1
python tools/level_times.py parse $O/lone/l_kernel_trace.csv 7 > $O/levels_lone_tree.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/f20 -o l -- python tools/level_times.py run 20 26 > /dev/null 2>&1
python tools/level_times.py parse $O/f20/l_kernel_trace.csv 15 > $O/levels_forest_20.txt 2>&1
rm -rf $O/stats $O/airfri $O/lone $O/f20
ls -la $O; cat $O/timed_region_launch_avg.txt $O/levels_lone_tree.txt $O/levels_forest_20.txt
