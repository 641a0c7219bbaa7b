#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02/prof
mkdir -p $O
rm -rf $O/enc_fp32
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/enc_fp32 -o t -- python3 $R/bench.py --workload sorl_enc --steps 6 --warmup 2 --no-cpu-baseline > $O/enc_fp32.json 2> $O/enc_fp32.err
