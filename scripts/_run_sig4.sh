#!/bin/bash
set -e
mkdir -p gpurun_out/r02
rm -f gpurun_out/r02/sig4.log
for rep in 1 2 3; do
for m in event signal signal3; do
  echo "sync $m" >> gpurun_out/r02/sig4.log
  PORL_BENCH_SUSTAINED=0 PORL_PIPE_SYNC=$m python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])" >> gpurun_out/r02/sig4.log
done
done
