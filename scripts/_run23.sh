mkdir -p gpurun_out/r02
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 50" "--steps 20 --warmup 5 --no-pipeline" "--steps 20 --warmup 50 --no-pipeline" "--steps 200 --warmup 5"; do
python3 bench.py --gpus 1 $args --no-cpu-baseline --no-roofline > gpurun_out/r02/b23.json 2> gpurun_out/r02/b23.err
python -c "
import json; d=json.loads(open('gpurun_out/r02/b23.json').read().strip().splitlines()[-1]); print('$args', round(d['value'],1), round(1e3*d['ms_per_step'],1))"
done
