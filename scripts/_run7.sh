set -e
mkdir -p gpurun_out/r02
python -m pytest tests/test_por_gpu.py tests/test_dp_gpu.py -m gpu -x -q > gpurun_out/r02/gpu_tests7.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests7.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests7.log
python scripts/bench_select_action.py > gpurun_out/r02/select_action.log 2>&1; cat gpurun_out/r02/select_action.log
python bench.py --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/r02/bench7.json 2> gpurun_out/r02/bench7.err; cat gpurun_out/r02/bench7.json
