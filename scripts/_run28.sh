set -e
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests28.log 2>&1 || { tail -50 gpurun_out/r02/gpu_tests28.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests28.log
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > gpurun_out/r02/bench28.json 2> gpurun_out/r02/bench28.err; python -c "
import json; d=json.load(open('gpurun_out/r02/bench28.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved']); print(d['roofline']['step_launches_us'])"
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --no-pipeline > gpurun_out/r02/bench28b.json 2> gpurun_out/r02/bench28b.err; cut -c1-200 gpurun_out/r02/bench28b.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r02/bench28c.json 2> gpurun_out/r02/bench28c.err; cut -c1-200 gpurun_out/r02/bench28c.json
