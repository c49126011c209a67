#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-60s %.3e' % (' '.join(sys.argv[1:]), d['value']))" "$@"; }
run --steps 64 --warmup 16 --streams 1 --plan 64
run --steps 128 --warmup 16 --streams 2 --plan 64,64
run --steps 128 --warmup 16 --streams 1 --plan 64,64
run --steps 256 --warmup 16 --streams 2 --plan 64,64,64,64
run --steps 256 --warmup 16 --streams 1 --plan 64,64,64,64
run --steps 32 --warmup 8 --streams 2 --plan 32
run --steps 32 --warmup 8 --streams 2 --plan 16,16
run --steps 16 --warmup 4 --streams 2 --plan 16
run --steps 16 --warmup 4 --streams 2 --plan 8,8
run --steps 8 --warmup 2 --streams 2 --plan 8
run --steps 8 --warmup 2 --streams 2 --plan 4,4
run --steps 96 --warmup 16 --streams 2 --plan 48,48
run --steps 96 --warmup 16 --streams 2 --plan 64,32
