# round 3, GPU run C: the double-steered inversion - micro-benchmark + check, pedersen parity tests,
# per-level traces (lone tree, forest of 20) and the default bench line
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03c
mkdir -p $O
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../stark-perpetual_amd/csrc inv_quad.hip -o inv_quad 2>/dev/null)
timeout 120 ./tools/ubench/inv_quad > $O/inv_quad.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_pedersen.py tests/test_gpu_state.py tests/test_gpu_ecdsa.py tests/test_gpu_keyed_verify.py -x -q 2>&1 | tail -5 > $O/pytest.txt
for T in 1 20; do
  rocprofv3 --kernel-trace --output-format csv -d $O/lt$T -o t -- python tools/level_times.py run $T 26 > $O/level_times_$T.txt 2>&1
  N=$([ $T = 1 ] && echo 16 || echo 20)
  python tools/level_times.py parse $O/lt$T/t_kernel_trace.csv $((2 * N)) > $O/levels_$T.txt 2>&1
done
rm -rf $O/lt1 $O/lt20
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(json.dumps(d["summary"], indent=0))
PY
cat $O/inv_quad.txt $O/pytest.txt
