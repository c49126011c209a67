# round 4, closing pass on the final binaries: the r04 profile set again (table build changed the kernel list), whole GPU
# suite, smoke, default bench line
export TMPDIR=/tmp
O=gpurun_out/r04l; mkdir -p $O
bash tools/run_r04_prof.sh > $O/prof.log 2>&1; tail -3 $O/prof.log
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python __graft_entry__.py smoke 2>/dev/null | tail -1
python bench.py > $O/bench_default.json 2> $O/bench.err; python -c "
import json;d=json.load(open('$O/bench_default.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['whole_region']['frac'],d['summary']['airfri_commits_per_sec'],d['summary']['single_tree_ms'])"
