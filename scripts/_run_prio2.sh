#!/bin/bash
set -e
mkdir -p gpurun_out/r02
rm -f gpurun_out/r02/prio2.log
for p in 0 -1; do
  for pad in "18432,0,1024" "18432,18432,1024"; do
  echo "side priority $p pad $pad" >> gpurun_out/r02/prio2.log
  PORL_BENCH_SUSTAINED=0 PORL_IQL_PAD=$pad PORL_SIDE_PRIORITY=$p python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])" >> gpurun_out/r02/prio2.log
  done
done
