# round 4: same-box A/B of the round-3 library (additive Montgomery reduction) against this round's (subtractive),
# because box-to-box spread (+- 5 %) is as large as the change; plus the RFC 6979 chain micro-benchmark and the
# tests that were fixed after pass d
export TMPDIR=/tmp
O=gpurun_out/r04e; mkdir -p $O
L=stark-perpetual_amd/lib/libstarkperp.so
cp $L /tmp/lib_current.so
for rep in 1 2; do
  for v in r03 r04; do
    cp tools/ab_libs/libstarkperp_$v.so $L
    echo "== $v (pass $rep)" >> $O/ab.txt
    python tools/quick_bulk.py 22 26 2>/dev/null >> $O/ab.txt
    python tools/level_times.py run 20 26 2>/dev/null >> $O/ab.txt
    python tools/level_times.py run 1 26 2>/dev/null >> $O/ab.txt
    python bench.py --no-cpu-baseline --no-extras --no-airfri 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench.py default: %.4g hashes/s, %.4f ms/step, bulk launch avg %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))" >> $O/ab.txt
  done
done
cp /tmp/lib_current.so $L
cat $O/ab.txt
tools/ubench/rfc6979_chain > $O/rfc6979_chain.txt 2>&1; cat $O/rfc6979_chain.txt
timeout 900 python -m pytest tests/test_program_hash.py tests/test_gpu_rccl.py tests/test_gpu_ecdsa.py -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
