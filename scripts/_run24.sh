mkdir -p gpurun_out/r02
for t in 3 1 0 2; do
PORL_L0_TILE=$t python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-pipeline > gpurun_out/r02/b24.json 2> gpurun_out/r02/b24.err
python -c "
import json; d=json.load(open('gpurun_out/r02/b24.json')); print('tile $t', round(d['value'],1), {k.split(':')[0]:v for k,v in d['roofline']['step_launches_us'].items() if 'L0fwd' in k})"
done
