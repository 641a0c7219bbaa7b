set -e
mkdir -p gpurun_out/r02/prof_a
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02/prof_a -o por -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-pipeline > $GRAFT_REPO_ROOT/gpurun_out/r02/prof_a/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02/prof_a/err.log || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/r02/prof_a/err.log; exit 1; }
cat $GRAFT_REPO_ROOT/gpurun_out/r02/prof_a/bench.json
find $GRAFT_REPO_ROOT/gpurun_out/r02/prof_a -name "*kernel_stats*" | head
