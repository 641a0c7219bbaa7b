mkdir -p gpurun_out/r02
python scripts/bench_overlap.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/overlap.log
