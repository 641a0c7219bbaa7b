set -e
mkdir -p gpurun_out/r02
python -m pytest tests/test_por_gpu.py -m gpu -x -q > gpurun_out/r02/gpu_tests5.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests5.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests5.log
python bench.py --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/r02/bench5.json 2> gpurun_out/r02/bench5.err; cat gpurun_out/r02/bench5.json
PORL_SIDE_PRIORITY=-1 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-roofline > gpurun_out/r02/bench5b.json 2> gpurun_out/r02/bench5b.err; cat gpurun_out/r02/bench5b.json
