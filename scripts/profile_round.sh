#!/bin/bash
# Profiles of one round (PORL_ROUND, default r03; run on the GPU box through gpurun).  Kernel traces and PMC passes are SEPARATE rocprofv3 runs
# (counters never together with a trace: MI355X_MICROARCH.md), the program stands directly after `--`.
set -e
R=$GRAFT_REPO_ROOT
RD=${PORL_ROUND:-r03}
O=$R/gpurun_out/$RD/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-secondary"
# 1. POR, every update back to back on one stream (each kernel alone on the chip): per-kernel durations
rocprofv3 --kernel-trace -d $O/por_serial -o t -- $B --no-pipeline > $O/por_serial.json 2> $O/por_serial.err
# 2. POR as benchmarked (policy phase pipelined on the side stream)
rocprofv3 --kernel-trace -d $O/por_pipelined -o t -- $B > $O/por_pipelined.json 2> $O/por_pipelined.err
# 3. HBM traffic and MFMA activity of the POR kernels (serial order), one counter group per pass
S="python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --no-pipeline"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o t -- $S > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o t -- $S > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES -d $O/pmc_mfma -o t -- $S > /dev/null 2> $O/pmc_mfma.err || \
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/pmc_mfma -o t -- $S > /dev/null 2> $O/pmc_mfma.err
# 4. secondary workloads: kernel traces
rocprofv3 --kernel-trace -d $O/cql -o t -- python3 $R/bench.py --workload cql --steps 200 --warmup 20 --no-roofline > $O/cql.json 2> $O/cql.err
rocprofv3 --kernel-trace -d $O/enc_fp32 -o t -- python3 $R/bench.py --workload sorl_enc --steps 6 --warmup 2 --no-cpu-baseline > $O/enc_fp32.json 2> $O/enc_fp32.err
rocprofv3 --kernel-trace -d $O/enc_bf16 -o t -- python3 $R/bench.py --workload sorl_enc --steps 6 --warmup 2 --no-cpu-baseline --enc-dtype bf16 > $O/enc_bf16.json 2> $O/enc_bf16.err
rocprofv3 --kernel-trace -d $O/enc_bf16_84 -o t -- python3 $R/bench.py --workload sorl_enc --steps 20 --warmup 3 --no-cpu-baseline --enc-dtype bf16 --angle-bins 84 --dist-bins 84 > $O/enc_bf16_84.json 2> $O/enc_bf16_84.err
ls -la $O $O/*/ | head -60
