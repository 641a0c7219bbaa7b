set -e
mkdir -p gpurun_out/r02
python -m pytest tests/test_por_gpu.py tests/test_dp_gpu.py -m gpu -x -q > gpurun_out/r02/gpu_tests2.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests2.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests2.log
python scripts/bench_host_cost.py > gpurun_out/r02/host_cost.log 2>&1; cat gpurun_out/r02/host_cost.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02/bench2.json 2> gpurun_out/r02/bench2.err; cat gpurun_out/r02/bench2.json
