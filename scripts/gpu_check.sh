cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rocminfo | grep -m2 -E 'gfx|Marketing' ; nproc; grep -m1 'model name' /proc/cpuinfo
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo smoke=$?; tail -5 gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest=$?; tail -15 gpurun_out/pytest_gpu.log
timeout 400 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1; echo bench=$?; tail -3 gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o ip1g -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo prof=$?; tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof.log
find $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -name '*stats*' | head; 
