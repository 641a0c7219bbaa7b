#!/bin/bash
set -e
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_por_gpu.py tests/test_dp_gpu.py -x -q -m gpu > gpurun_out/r02/f1_tests.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r02/f1_driver.json 2> gpurun_out/r02/f1.err
python bench.py > gpurun_out/r02/f1_default.json 2>> gpurun_out/r02/f1.err
PORL_BENCH_SUSTAINED=0 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-roofline > gpurun_out/r02/f1_1000.json 2>> gpurun_out/r02/f1.err
