# round 4, final pass: whole GPU suite + smoke on the final binaries, default bench line, w = 21 A/B of the two libraries
export TMPDIR=/tmp
O=gpurun_out/r04g; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1; tail -20 $O/pytest.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench.err; python -c "
import json;d=json.load(open('$O/bench_default.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['whole_region']['frac'],d['summary']['airfri_commits_per_sec'])"
L=stark-perpetual_amd/lib/libstarkperp.so; cp $L /tmp/lib_current.so
for v in r03 r04; do cp tools/ab_libs/libstarkperp_$v.so $L; echo "== $v, 21-bit windows" >> $O/ab21.txt; python tools/quick_bulk.py 22 21 2>/dev/null >> $O/ab21.txt; done
cp /tmp/lib_current.so $L; cat $O/ab21.txt
