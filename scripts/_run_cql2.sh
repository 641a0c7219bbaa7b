#!/bin/bash
set -e
mkdir -p gpurun_out/r02
timeout -k 10 500 python -m pytest tests/test_cql_gpu.py -x -q -m gpu > gpurun_out/r02/cql2_tests.log 2>&1
python bench.py --workload cql --steps 500 --warmup 50 > gpurun_out/r02/cql2_bench.json 2> gpurun_out/r02/cql2_bench.err
python scripts/bench_cql_prof.py > gpurun_out/r02/cql2_stamps.log 2>&1 || true
