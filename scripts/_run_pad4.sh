#!/bin/bash
set -e
mkdir -p gpurun_out/r02
rm -f gpurun_out/r02/pad4.log
run() {
  echo "map=$1 pad=$2" >> gpurun_out/r02/pad4.log
  PORL_BENCH_SUSTAINED=0 PORL_TILE_MAP=$1 PORL_GEMM_LDS_PAD=$2 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print(d['value']); print({k:v for k,v in d['roofline']['step_launches_us'].items() if 'V3' in k or 'V5' in k or 'P2' in k or 'P6' in k})" >> gpurun_out/r02/pad4.log
}
run "3,3,3,3" 0
run "3,3,3,3" 18432
run "" 0
