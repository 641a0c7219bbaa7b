#!/bin/bash
# Counter passes for the secondary workloads (config 3 CQL; config 5 encoder only with ENC=1: under --pmc its
# forward takes minutes per pass and prints nothing meanwhile — keep a progress file going when enabling it)
# Original intent: secondary workloads (config 3 CQL, config 5 encoder): HBM bytes and matrix-pipe activity per launch.
# Separate rocprofv3 --pmc runs (never with a trace), program directly after `--`; summaries are written on the box.
set -e
R=$GRAFT_REPO_ROOT
RD=${PORL_ROUND:-r03}
O=$R/gpurun_out/$RD/prof2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="python3 $R/bench.py --workload cql --steps 40 --warmup 5 --no-cpu-baseline"
S="python3 $R/bench.py --workload sorl_enc --batch 256 --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE -d $O/cql_fetch -o t -- $C > /dev/null 2> $O/cql_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/cql_write -o t -- $C > /dev/null 2> $O/cql_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES -d $O/cql_mfma -o t -- $C > /dev/null 2> $O/cql_mfma.err
if [ "$ENC" = "1" ]; then
rocprofv3 --pmc FETCH_SIZE -d $O/enc_fetch -o t -- $S > /dev/null 2> $O/enc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/enc_write -o t -- $S > /dev/null 2> $O/enc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES -d $O/enc_mfma -o t -- $S > /dev/null 2> $O/enc_mfma.err
fi
cd $R
python3 scripts/make_counters_secondary.py $O $R/gpurun_out/$RD/profiles_out
rm -rf $O/*/
