#!/bin/bash
set -e
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r02/verify_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02/verify_smoke.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r02/verify_bench_driver.json 2> gpurun_out/r02/verify_bench.err
python bench.py --workload cql > gpurun_out/r02/verify_cql.json 2>> gpurun_out/r02/verify_bench.err
python bench.py --workload sorl_enc --steps 6 --warmup 2 --enc-dtype bf16 --no-cpu-baseline > gpurun_out/r02/verify_enc_bf16.json 2>> gpurun_out/r02/verify_bench.err
PORL_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02/verify_gloo2.json 2>> gpurun_out/r02/verify_bench.err
