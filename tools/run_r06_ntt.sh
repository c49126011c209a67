#!/bin/bash
# what one pass of ntt_tile_kernel costs with ZERO stages (the kernel's own data path: its HBM
# ceiling) against the real passes, and the cheap variants of the data path.  On the GPU box: bash tools/run_r06_ntt.sh
# Variant libraries (built in the container, csrc/build_<v>/): see the loop below.
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"
O=$R/gpurun_out/r06ntt
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
OUT=$O/r06_ntt_pass_ceiling.txt
: > $OUT
run() {  # label, env assignments...
  local label="$1"; shift
  rm -rf $O/tr
  env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python $R/tools/quick_lde.py > $O/last.txt 2> $O/last.err
  local T=$(find $O/tr -name "*kernel_trace.csv" | head -1)
  python $R/tools/ntt_passes.py $T "$label" ${PER:-5} >> $OUT
  grep "lde 4 cols" $O/last.txt | sed "s/^/                                   host-timed, traced run: /" >> $OUT
}
run "product"                       STARKPERP_X=0
run "product, zero stages (copy)"   STARKPERP_NTT_PROBE=copy
for v in ${NTT_VARIANTS:-}; do
  lib=$R/stark-perpetual_amd/csrc/build_$v/libstarkperp_$v.so
  [ -f $lib ] || { echo "missing $lib" >> $OUT; continue; }
  case $v in t10*|t9*) PER=6;; *) PER=5;; esac
  run "$v"                          STARKPERP_LIB=$lib
  run "$v, zero stages (copy)"      STARKPERP_LIB=$lib STARKPERP_NTT_PROBE=copy
  PER=5
done
# untraced host timing of the product library
python $R/tools/quick_lde.py >> $OUT 2>> $O/last.err
rm -rf $O/tr
cat $OUT
