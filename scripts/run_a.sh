R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_por_gpu.py tests/test_dp_gpu.py tests/test_dataloader_gpu.py -x -q -m gpu > $O/oc_test.log 2>&1 || { tail -30 $O/oc_test.log; exit 1; }
tail -2 $O/oc_test.log
echo "== one call"; python scripts/bench_small_cfg.py 2>&1 | grep pipeline=1
echo "== phase calls"; PORL_PIPE_ONECALL=0 python scripts/bench_small_cfg.py 2>&1 | grep pipeline=1
for m in 1 0 1 0; do
  PORL_PIPE_ONECALL=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/oc.json 2>> $O/oc.err
  python -c "import json,sys; d=json.loads(open('$O/oc.json').read().strip().splitlines()[-1]); print('onecall=$m 20-step', round(d['value'],1), 'sustained', round(d['sustained_1000_updates_per_sec'],1))"
done
