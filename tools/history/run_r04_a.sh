export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
bash tools/run_r04_valu.sh > $O/valu.log 2>&1
python tools/quick_bulk.py 22 26 > $O/bulk26.txt 2>&1
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
cat $O/bulk26.txt
