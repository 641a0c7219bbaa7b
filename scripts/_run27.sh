mkdir -p gpurun_out/r02
for t in "3,3,2,3" "3,3,3,3"; do for pr in 0 -1; do
PORL_SIDE_PRIORITY=$pr PORL_TILE_MAP=$t python bench.py --steps 500 --warmup 30 --no-cpu-baseline --no-roofline > gpurun_out/r02/b27.json 2> gpurun_out/r02/b27.err
python -c "
import json; d=json.load(open('gpurun_out/r02/b27.json')); print('map [$t] prio $pr', round(d['value'],1))"
done; done
PORL_TILE_MAP="3,3,2,3" python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r02/b27.json 2> gpurun_out/r02/b27.err
python -c "
import json; d=json.load(open('gpurun_out/r02/b27.json')); print('driver form', round(d['value'],1))"
