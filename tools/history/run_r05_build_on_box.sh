# build provenance on the GPU box - make clean, __graft_entry__.build() with the box's own hipcc,
# smoke() on cuda:0, the sha256 of what was built, then the GPU suite and the driver's bench command on THAT binary.
O=gpurun_out/r05box; mkdir -p $O
{
echo "== $(date -u +%Y-%m-%dT%H:%M:%SZ) on $(hostname): $(nproc) host threads"; hipcc --version | head -2
echo "== shipped library (built in the build container): $(sha256sum stark-perpetual_amd/lib/libstarkperp.so)"
make -C stark-perpetual_amd/csrc clean > /dev/null
rm -rf oracle/_build tests/host/host_shim.so
echo "== make clean done; __graft_entry__.build():"
( time python -c "import __graft_entry__ as g; g.build()" ) 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *| \|generated when" | tail -12
echo "== built on this box: $(sha256sum stark-perpetual_amd/lib/libstarkperp.so)"
python -c "
import sys; sys.path.insert(0,'stark-perpetual_amd')
from starkperp import _lib; print(_lib.load().sp_build_info().decode())"
echo "== smoke():"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
} > $O/build_on_gpu_box.txt 2>&1
cat $O/build_on_gpu_box.txt
( time python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
echo "== GPU suite on the box-built library: $(grep -E "passed|failed" $O/pytest_gpu.txt | tail -1 | grep -E 'passed|failed')" >> $O/build_on_gpu_box.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?
python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); s=d['summary']
print({k:s[k] for k in ('pedersen_hashes_per_sec','roofline_frac_bulk_launches','roofline_frac_at_held_clock','roofline_frac_whole_region','sclk_mhz_median','power_w_median','timed_total_s','burst_pedersen_hashes_per_sec','airfri_commits_per_sec','bulk_pedersen_hashes_per_sec','lib_sha256_16','parity_in_run')})"
