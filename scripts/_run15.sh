set -e
mkdir -p gpurun_out/r02/prof_b
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r02/prof_b -o por -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r02/prof_b/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02/prof_b/err.log || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/r02/prof_b/err.log; exit 1; }
cat $GRAFT_REPO_ROOT/gpurun_out/r02/prof_b/bench.json | cut -c1-300
