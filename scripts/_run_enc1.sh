#!/bin/bash
set -e
mkdir -p gpurun_out/r02
VARIANTS=plain,apro+resid,resid python scripts/bench_gemm_enc.py > gpurun_out/r02/gemm_enc2.log 2>&1
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py tests/test_fasternet_gpu.py tests/test_por_gpu.py -x -q -m gpu > gpurun_out/r02/enc1_tests.log 2>&1
python bench.py --workload sorl_enc --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02/enc1_bench.json 2> gpurun_out/r02/enc1_bench.err
python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/r02/enc1_por.json 2>> gpurun_out/r02/enc1_bench.err
