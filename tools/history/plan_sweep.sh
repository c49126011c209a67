#!/bin/bash
# Call-size schedules for the K = 64 timed steps of bench.py (run on the GPU box).
run() { python bench.py --no-cpu-baseline --no-extras --steps 64 --warmup 16 "$@" 2>/dev/null | tail -1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-46s %.3e' % (' '.join(sys.argv[1:]), d['value']))" "$@"; }
run --streams 2
run --streams 2 --plan 40,24
run --streams 2 --plan 48,16
run --streams 2 --plan 32,16,16
run --streams 2 --plan 24,24,16
run --streams 2 --plan 32,16,8,8
run --streams 2 --plan 24,16,12,8,4
run --streams 3 --plan 24,16,12,8,4
run --streams 3 --plan 32,20,12
run --streams 3 --plan 28,20,16
run --streams 4 --plan 16,16,16,16
run --streams 4 --plan 28,18,12,6
run --streams 2 --plan 16,16,16,16
run --streams 2 --plan 56,8
run --streams 2 --plan 64
