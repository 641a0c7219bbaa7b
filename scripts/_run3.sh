set -e
mkdir -p gpurun_out/r02
python -m pytest tests/test_por_gpu.py tests/test_dp_gpu.py tests/test_native_abi.py -m gpu -x -q > gpurun_out/r02/gpu_tests3.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests3.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests3.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02/bench3.json 2> gpurun_out/r02/bench3.err; cat gpurun_out/r02/bench3.json
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-roofline > gpurun_out/r02/bench3b.json 2> gpurun_out/r02/bench3b.err; cat gpurun_out/r02/bench3b.json
