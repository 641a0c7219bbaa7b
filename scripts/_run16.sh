set -e
mkdir -p gpurun_out/r02
python -m pytest tests/test_por_gpu.py tests/test_dp_gpu.py -m gpu -x -q > gpurun_out/r02/gpu_tests16.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests16.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests16.log
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-roofline > gpurun_out/r02/bench16.json 2> gpurun_out/r02/bench16.err; cut -c1-260 gpurun_out/r02/bench16.json
