mkdir -p gpurun_out/r02
for sp in 0 50 300; do for rep in 1 2; do
PORL_BENCH_SPINUP_MS=$sp python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r02/b26.json 2> gpurun_out/r02/b26.err
python -c "
import json; d=json.loads(open('gpurun_out/r02/b26.json').read().strip().splitlines()[-1]); print('spinup $sp ms:', round(d['value'],1), round(1e3*d['ms_per_step'],1))"
done; done
