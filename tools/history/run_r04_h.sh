# round 4: the doubling-schedule table build - time of sp_init under both builds, the parity suites that touch the tables
export TMPDIR=/tmp
O=gpurun_out/r04h; mkdir -p $O
(python tools/quick_init.py 26; STARKPERP_TABLE_BUILD=direct python tools/quick_init.py 26; python tools/quick_init.py 21; STARKPERP_TABLE_BUILD=direct python tools/quick_init.py 21) 2>/dev/null > $O/table_build.txt; cat $O/table_build.txt
timeout 1500 python -m pytest tests/test_gpu_pedersen.py tests/test_gpu_window_plans.py tests/test_gpu_ecdsa.py tests/test_gpu_keyed_verify.py -m gpu -q -x > $O/pytest.log 2>&1; tail -6 $O/pytest.log
python tools/quick_bulk.py 22 26 2>/dev/null | tail -1
STARKPERP_WINDOW_BITS=26 timeout 600 python tools/soak_sizes.py 2>/dev/null | tail -2
