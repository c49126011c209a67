export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
for t in tests/test_program_hash.py tests/test_gpu_rccl.py; do
  timeout 420 python -u -m pytest $t -m gpu -v -x -o faulthandler_timeout=200 > $O/$(basename $t).log 2>&1
  echo "rc=$? $t"; tail -25 $O/$(basename $t).log
done
