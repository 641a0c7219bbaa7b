mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests14.log 2>&1 || { tail -50 gpurun_out/r02/gpu_tests14.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests14.log
