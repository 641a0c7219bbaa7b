#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02/prof
cd $R
timeout -k 10 900 python -m pytest tests/test_por_gpu.py tests/test_dp_gpu.py tests/test_ops.py -x -q -m gpu > gpurun_out/r02/sig2_tests.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02/sig2_driver.json 2> gpurun_out/r02/sig2.err
rm -rf $R/gpurun_out/r02/prof/por_pipelined
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/r02/prof/por_pipelined -o t -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline > $R/gpurun_out/r02/prof/por_pipelined.json 2> $R/gpurun_out/r02/prof/por_pipelined.err
