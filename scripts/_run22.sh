mkdir -p gpurun_out/r02
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r02/bench_driver_form.json 2> gpurun_out/r02/bench_driver_form.err
python -c "
import json; d=json.loads(open('gpurun_out/r02/bench_driver_form.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['cpu_baseline'], d.get('speedup_vs_cpu_baseline'))"
tail -4 gpurun_out/r02/bench_driver_form.err
