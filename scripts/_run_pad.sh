#!/bin/bash
set -e
mkdir -p gpurun_out/r02
rm -f gpurun_out/r02/pad.log
for p in 0 18432 45056; do
  echo "lds pad $p" >> gpurun_out/r02/pad.log
  PORL_BENCH_SUSTAINED=0 PORL_GEMM_LDS_PAD=$p python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])" >> gpurun_out/r02/pad.log
done
