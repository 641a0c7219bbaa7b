mkdir -p gpurun_out/r02
python -m pytest tests/test_fasternet_gpu.py -m gpu -x -q -k "bf16" > gpurun_out/r02/gpu_tests19.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests19.log; exit 1; }
tail -2 gpurun_out/r02/gpu_tests19.log
python bench.py --workload sorl_enc --steps 10 --warmup 3 --no-cpu-baseline --enc-dtype bf16 > gpurun_out/r02/bench_enc_bf16b.json 2> gpurun_out/r02/bench_enc_bf16b.err; cut -c1-1700 gpurun_out/r02/bench_enc_bf16b.json
