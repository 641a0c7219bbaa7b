mkdir -p gpurun_out/r02
python -m pytest tests/test_cql_gpu.py tests/test_per_gpu.py -m gpu -x -q > gpurun_out/r02/gpu_tests11.log 2>&1 || { tail -60 gpurun_out/r02/gpu_tests11.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests11.log
python bench.py --workload sorl_enc --steps 10 --warmup 3 > gpurun_out/r02/bench_enc0.json 2> gpurun_out/r02/bench_enc0.err; cat gpurun_out/r02/bench_enc0.json
