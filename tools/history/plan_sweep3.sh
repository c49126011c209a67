#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-60s %.3e' % (' '.join(sys.argv[1:]), d['value']))" "$@"; }
run --steps 128 --warmup 16 --streams 2 --plan 128
run --steps 128 --warmup 16 --streams 2 --plan 64,64
run --steps 128 --warmup 16 --streams 4 --plan 32,32,32,32
run --steps 128 --warmup 16 --streams 3 --plan 48,40,40
run --steps 256 --warmup 16 --streams 2 --plan 128,128
run --steps 256 --warmup 16 --streams 2 --plan 256
run --steps 512 --warmup 16 --streams 2 --plan 256,256
run --steps 512 --warmup 16 --streams 2 --plan 128,128,128,128
