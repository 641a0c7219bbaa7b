mkdir -p gpurun_out/r02
python scripts/bench_cql_prof.py > gpurun_out/r02/cql_prof.log 2>&1; cat gpurun_out/r02/cql_prof.log
