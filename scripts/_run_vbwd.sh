#!/bin/bash
set -e
mkdir -p gpurun_out/r02
rm -f gpurun_out/r02/vbwd.log
for t in -1 2 1 0; do
  echo "vbwd tile $t" >> gpurun_out/r02/vbwd.log
  for i in 1 2; do
  PORL_VBWD_TILE=$t python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])" >> gpurun_out/r02/vbwd.log
  done
done
