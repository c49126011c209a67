# round 3, GPU run D: NTT kernel (register-blocked radix-8) - parity tests and the prover phases
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03d
mkdir -p $O
timeout 300 python tools/quick_ntt_check.py 2>&1 | tail -3 > $O/ntt_check.txt; timeout 1200 python -m pytest tests/test_gpu_stark.py tests/test_gpu_builtins.py -x -q 2>&1 | tail -5 > $O/pytest.txt
python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
a = d["airfri"]
print("commits/s", a.get("commits_per_sec"), "s/job", a.get("seconds_per_job"))
for k, v in a["phases"].items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("ms", "hbm_frac", "frac_of_8TBs", "algorithmic_bytes")} if isinstance(v, dict) else v)
PY
cat $O/ntt_check.txt $O/pytest.txt
