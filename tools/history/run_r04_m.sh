export TMPDIR=/tmp
O=gpurun_out/r04m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ecdsa.py -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
(echo "== compacted from 131072 items on (default)"; python tools/quick_sign.py 2>/dev/null; echo "== one-kernel signer (STARKPERP_SIGN_COMPACT_MIN=0)"; STARKPERP_SIGN_COMPACT_MIN=0 python tools/quick_sign.py 2>/dev/null; echo "== compacted at every size (STARKPERP_SIGN_COMPACT_MIN=1)"; STARKPERP_SIGN_COMPACT_MIN=1 python tools/quick_sign.py 2>/dev/null) | cut -c1-75 > $O/sign.txt; cat $O/sign.txt
