set -e
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests8.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests8.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests8.log
PORL_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 30 --warmup 5 --rows-per-gpu 100000 --no-roofline > gpurun_out/r02/bench_gloo2b.json 2> gpurun_out/r02/bench_gloo2b.err; cat gpurun_out/r02/bench_gloo2b.json
