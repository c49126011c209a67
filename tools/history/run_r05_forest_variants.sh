#!/bin/bash
# the mid levels of the driver's 20-tree forest (levels 4 - 8: 79 360 hashes in 265 us).  Every
# variant that the library's switches can express, each in its own process on ONE box, sustained (400 builds warm-up,
# 400 timed) so that the numbers are comparable; plus the hipGraph replay of the baseline.
#   bash tools/run_r05_forest_variants.sh > gpurun_out/r05/forest_variants.txt
cd "$(dirname "$0")/.."
run() { echo "== $*"; env LEVEL_TIMES_SUSTAINED=1 "$@" python tools/level_times.py run 20 26 2>&1 | grep -v amdgpu.ids; }
run A=baseline
run LEVEL_TIMES_GRAPH=1
run STARKPERP_SPLIT_LANES=131072      # level 4 on 2 lanes per hash (81 920 lanes = 1.25 waves per SIMD), 5 on 4, 6 on 8
run STARKPERP_SPLIT_LANES=98304       # level 4 stays on 1 lane, level 5 (20 480) on 4 lanes, level 6 on 8
run STARKPERP_SPLIT_LANES=32768       # fewer lanes: level 5 on 1 lane, 6 on 2, 7 on 4 ...
run STARKPERP_QUAD_MAX=4096           # quad kernels one level earlier: 8 quads up to 4096 hashes, top kernel from 4096 down
run STARKPERP_QUAD_MAX=1024           # ... one level later
run STARKPERP_NO_TOP_FUSION=1         # the small levels one launch each (what the four-levels-per-launch kernel buys)
run STARKPERP_NO_LEVEL_SPLIT=1        # levels 2 and 3 without the whole-rounds + lane-split remainder cut
run STARKPERP_NO_FUSE=1               # levels 4 - 6 with a separate finish launch instead of the fused inversion
