set -e
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests1.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests1.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests1.log
python scripts/bench_two_streams.py > gpurun_out/r02/two_streams.log 2>&1; cat gpurun_out/r02/two_streams.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02/bench_base.json 2> gpurun_out/r02/bench_base.err; cat gpurun_out/r02/bench_base.json
PORL_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 10 --rows-per-gpu 100000 > gpurun_out/r02/bench_gloo2.json 2> gpurun_out/r02/bench_gloo2.err; cat gpurun_out/r02/bench_gloo2.json; tail -5 gpurun_out/r02/bench_gloo2.err
