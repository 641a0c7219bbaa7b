#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02/prof
mkdir -p $O
rm -rf $O/por_pipelined
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/por_pipelined -o t -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline > $O/por_pipelined.json 2> $O/por_pipelined.err
ls -la $O/por_pipelined
