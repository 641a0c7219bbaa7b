mkdir -p gpurun_out/r02
python -m pytest tests/test_fasternet_gpu.py -m gpu -x -q > gpurun_out/r02/gpu_tests18.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests18.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests18.log
python bench.py --workload sorl_enc --steps 10 --warmup 3 --no-cpu-baseline --enc-dtype bf16 > gpurun_out/r02/bench_enc_bf16.json 2> gpurun_out/r02/bench_enc_bf16.err; cut -c1-2600 gpurun_out/r02/bench_enc_bf16.json
python bench.py --workload sorl_enc --steps 10 --warmup 3 --no-cpu-baseline --enc-dtype bf16 --angle-bins 84 --dist-bins 84 > gpurun_out/r02/bench_enc_bf16_84.json 2> gpurun_out/r02/bench_enc_bf16_84.err; cut -c1-600 gpurun_out/r02/bench_enc_bf16_84.json
