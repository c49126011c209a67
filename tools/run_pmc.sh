# rocprofv3 passes behind profiles/r02_*: kernel stats of the driver's command, PMC traffic of the same
# command (separate FETCH_SIZE / WRITE_SIZE passes), SQ_INSTS_VALU of the bulk kernels.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02prof
mkdir -p $O
CMD="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-airfri"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- $CMD > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o b -- $CMD > $O/fetch_bench.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o b -- $CMD > $O/write_bench.json 2> $O/write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -o b -- python tools/bulk_only.py 22 26 > $O/sq.log 2>&1
python tools/valu_counts.py $O/sq/b_counter_collection.csv $O/sq/b_kernel_trace.csv 22 26 > $O/valu_counts_w26.json
python tools/trace_timed_avg.py $O/stats/b_kernel_trace.csv 4 > $O/timed_region_launch_avg.txt 2>&1
# airfri workload stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/airfri -o a -- python bench.py --workload airfri --steps 6 --warmup 3 --no-cpu-baseline > $O/airfri_bench.json 2> $O/airfri.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/afetch -o a -- python bench.py --workload airfri --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/awrite -o a -- python bench.py --workload airfri --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
ls $O $O/stats | head -30
# keep only the summaries (the raw traces are large)
rm -f $O/*/b_kernel_trace.csv.bak
du -sh $O
