#!/bin/bash
set -e
mkdir -p gpurun_out/r02
rm -f gpurun_out/r02/spin2.log
for ms in 50 200 600; do
  for i in 1 2; do
  echo "spin $ms" >> gpurun_out/r02/spin2.log
  PORL_BENCH_SUSTAINED=0 PORL_BENCH_SPINUP_MS=$ms python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])" >> gpurun_out/r02/spin2.log
  done
done
