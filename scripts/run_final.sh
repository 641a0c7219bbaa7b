set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1 || { tail -30 $O/gputest.log; exit 1; }
tail -1 $O/gputest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash scripts/run_prof.sh > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
tail -18 $O/prof.log
