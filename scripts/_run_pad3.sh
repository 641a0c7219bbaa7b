#!/bin/bash
set -e
mkdir -p gpurun_out/r02
rm -f gpurun_out/r02/pad3.log
run() {
  echo "iql_pad=$1" >> gpurun_out/r02/pad3.log
  PORL_BENCH_SUSTAINED=0 PORL_IQL_PAD=$1 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])" >> gpurun_out/r02/pad3.log
}
run "18432,18432"
run "18432,0"
run "0,18432"
run "18432,18432,500"
run "18432,18432,300"
run "24576,24576"
run "18432,18432,1000"
