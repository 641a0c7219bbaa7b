mkdir -p gpurun_out/r02
python -m pytest tests/test_cql_gpu.py tests/test_per_gpu.py tests/test_ops.py -m gpu -x -q > gpurun_out/r02/gpu_tests10.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests10.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests10.log
python scripts/bench_cql_prof.py > gpurun_out/r02/cql_prof2.log 2>&1; cat gpurun_out/r02/cql_prof2.log
