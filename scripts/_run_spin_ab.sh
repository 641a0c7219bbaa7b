#!/bin/bash
set -e
mkdir -p gpurun_out/r02
for i in 1 2 3; do
  PORL_BENCH_SPINUP_MS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline >> gpurun_out/r02/spin_ab0.log 2>/dev/null
  PORL_BENCH_SPINUP_MS=50 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline >> gpurun_out/r02/spin_ab50.log 2>/dev/null
done
