mkdir -p gpurun_out/r02
python -m pytest tests/test_fasternet_gpu.py tests/test_costmap_gpu.py tests/test_gemm_gpu.py -m gpu -x -q > gpurun_out/r02/gpu_tests12.log 2>&1 || { tail -60 gpurun_out/r02/gpu_tests12.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests12.log
python bench.py --workload sorl_enc --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02/bench_enc1.json 2> gpurun_out/r02/bench_enc1.err; cat gpurun_out/r02/bench_enc1.json
