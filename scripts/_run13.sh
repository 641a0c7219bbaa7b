mkdir -p gpurun_out/r02
python -m pytest tests/test_dist_gpu.py tests/test_cql_gpu.py -m gpu -x -q > gpurun_out/r02/gpu_tests13.log 2>&1 || { tail -50 gpurun_out/r02/gpu_tests13.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests13.log
rocprofv3 -L > gpurun_out/r02/counters.txt 2>&1 || true
grep -i -c "mfma" gpurun_out/r02/counters.txt
