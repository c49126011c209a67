#!/bin/bash
# round 3: re-tune the level plan after the inversion became 3x cheaper (finish lanes, quad / split thresholds)
run() { env "$@" python bench.py --no-cpu-baseline --no-extras --no-airfri --steps 20 --warmup 5 2>/dev/null | tail -1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-70s 20 trees %.3e' % (' '.join(sys.argv[1:]), d['value']))" "$@"
  env "$@" python bench.py --no-cpu-baseline --no-extras --no-airfri --steps 1 --warmup 2 2>/dev/null | tail -1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-70s lone tree %.4f ms' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@"; }
run X=0
run STARKPERP_FINISH_LANES=131072
run STARKPERP_FINISH_LANES=262144
run STARKPERP_QUAD_MAX=1024
run STARKPERP_QUAD_MAX=4096
run STARKPERP_SPLIT_LANES=131072
run STARKPERP_SPLIT_LANES=131072 STARKPERP_FINISH_LANES=131072
run STARKPERP_NO_QUAD2=1
run STARKPERP_NO_LEVEL_SPLIT=1
