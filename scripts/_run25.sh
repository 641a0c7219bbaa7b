set -e
mkdir -p gpurun_out/r02
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r02/smoke.log 2>&1 || { tail -20 gpurun_out/r02/smoke.log; exit 1; }
tail -2 gpurun_out/r02/smoke.log
python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_final.log 2>&1 || { tail -50 gpurun_out/r02/gpu_tests_final.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests_final.log
python bench.py > gpurun_out/r02/bench_final.json 2> gpurun_out/r02/bench_final.err; python -c "
import json; d=json.loads(open('gpurun_out/r02/bench_final.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['cpu_baseline'])"
python bench.py --workload cql > gpurun_out/r02/bench_cql_final.json 2> gpurun_out/r02/bench_cql_final.err; cut -c1-1200 gpurun_out/r02/bench_cql_final.json
