set -e
mkdir -p gpurun_out/r02
python -m pytest tests/test_por_gpu.py tests/test_dp_gpu.py tests/test_cql_gpu.py tests/test_dist_gpu.py -m gpu -x -q > gpurun_out/r02/gpu_tests20.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests20.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests20.log
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > gpurun_out/r02/bench20.json 2> gpurun_out/r02/bench20.err; python -c "
import json; d=json.load(open('gpurun_out/r02/bench20.json')); print(d['value'], d['ms_per_step']); print(d['roofline']['step_launches_us'])"
