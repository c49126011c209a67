# round 3, GPU run H: soak of the end-of-round binaries (double-steered inversions everywhere) + the Pedersen /
# state suites under 26-bit windows
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03h
mkdir -p $O
STARKPERP_WINDOW_BITS=26 timeout 1200 python -m pytest tests/test_gpu_pedersen.py tests/test_gpu_state.py tests/test_gpu_cabi.py -q 2>&1 | tail -3 > $O/pytest_w26.txt
timeout 1500 python tools/soak.py 22 131072 > $O/soak.txt 2>&1
timeout 900 python tools/soak_sizes.py > $O/soak_sizes.txt 2>&1
cat $O/pytest_w26.txt; tail -8 $O/soak.txt; tail -5 $O/soak_sizes.txt
