mkdir -p gpurun_out/r02
for f in 1 0; do
PORL_IQL_FOLD=$f python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-pipeline > gpurun_out/r02/bench21_$f.json 2> gpurun_out/r02/bench21_$f.err; python -c "
import json; d=json.load(open('gpurun_out/r02/bench21_$f.json')); print('fold=$f', d['value'], d['ms_per_step']); print({k:v for k,v in d['roofline']['step_launches_us'].items() if 'adam' in k or 'reduce' in k})"
done
