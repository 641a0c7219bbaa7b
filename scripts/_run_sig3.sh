#!/bin/bash
set -e
mkdir -p gpurun_out/r02
rm -f gpurun_out/r02/sig3.log
timeout -k 10 600 python -m pytest tests/test_por_gpu.py -x -q -m gpu > gpurun_out/r02/sig3_tests.log 2>&1
for m in signal event signal; do
  echo "sync $m" >> gpurun_out/r02/sig3.log
  PORL_BENCH_SUSTAINED=0 PORL_PIPE_SYNC=$m python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['final_losses'])" >> gpurun_out/r02/sig3.log
done
