#!/bin/bash
set -e
mkdir -p gpurun_out/r02
rm -f gpurun_out/r02/sweep5.log
run() {
  echo "pad=$1 l0=$2" >> gpurun_out/r02/sweep5.log
  PORL_BENCH_SUSTAINED=0 PORL_IQL_PAD=$1 PORL_L0_TILE=$2 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])" >> gpurun_out/r02/sweep5.log
}
run "18432,0,1024,0" ""
run "18432,0,1024,128" ""
run "18432,0,1024,0" "2"
run "18432,0,1024,128" "2"
run "18432,0,1024,0" ""
run "18432,0,1024,128" ""
