#!/bin/bash
# usage: tools/sweep.sh  (dev aid) - bench value for combinations of HW queues and streams
for QS in "16 16" "24 24" "32 32" "32 64" "64 64"; do set -- $QS
  echo -n "queues=$1 streams=$2: "
  GPU_MAX_HW_QUEUES=$1 python bench.py --no-cpu-baseline --no-extras --streams $2 --steps 128 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
done
