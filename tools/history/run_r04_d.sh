# round 4, fourth GPU pass: the whole GPU suite on the final binaries, the default bench line, soaks under both window plans
export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench.err; tail -c 600 $O/bench_default.json
python bench.py --force-dist --no-cpu-baseline --no-extras > $O/bench_force_dist.json 2> $O/bench_fd.err; tail -c 300 $O/bench_force_dist.json
(STARKPERP_WINDOW_BITS=26 timeout 900 python tools/soak.py 22 65536; STARKPERP_WINDOW_BITS=21 timeout 900 python tools/soak.py 21 32768; STARKPERP_WINDOW_BITS=26 timeout 600 python tools/soak_sizes.py; timeout 600 python tools/soak_trees.py) > $O/soak.txt 2>&1; tail -12 $O/soak.txt
