#!/bin/bash
set -e
mkdir -p gpurun_out/r02
python -c "import torch; print(torch.cuda.Stream.priority_range())" > gpurun_out/r02/prio.log 2>&1
for p in 1 0 -1; do
  echo "side priority $p" >> gpurun_out/r02/prio.log
  PORL_SIDE_PRIORITY=$p python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])" >> gpurun_out/r02/prio.log
done
