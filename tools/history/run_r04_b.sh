export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
bash tools/run_r04_valu.sh > $O/valu.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_rccl.py tests/test_program_hash.py tests/test_gpu_keyed_verify.py "tests/test_gpu_ecdsa.py::test_messages" -m gpu -x -q > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
timeout 1500 python -m pytest tests/test_gpu_window_plans.py -m gpu -x -q --durations=8 > $O/pytest_w.log 2>&1; tail -20 $O/pytest_w.log
