#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02/prof
mkdir -p $O
rm -rf $O/cql
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/cql -o t -- python3 $R/bench.py --workload cql --steps 200 --warmup 20 --no-roofline > $O/cql.json 2> $O/cql.err
ls -la $O/cql
