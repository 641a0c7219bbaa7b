#!/bin/bash
set -e
mkdir -p gpurun_out/r02
rm -f gpurun_out/r02/pad2.log
run() {
  echo "map=$1 pad=$2" >> gpurun_out/r02/pad2.log
  PORL_BENCH_SUSTAINED=0 PORL_TILE_MAP=$1 PORL_GEMM_LDS_PAD=$2 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])" >> gpurun_out/r02/pad2.log
}
run "3,3,3,3" 18432
run "3,3,3,3" 12288
run "2,2,2,2" 30000
run "0,0,0,0" 0
run "1,1,1,1" 30000
run "2,3,2,3" 18432
run "0,3,0,3" 18432
