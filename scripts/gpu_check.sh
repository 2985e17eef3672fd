# Round GPU check: parity tests, bench lines, rocprofv3 kernel stats.  Run on the GPU box via gpurun.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest=$?; tail -4 gpurun_out/pytest_gpu.log
timeout 400 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1; echo bench=$?; tail -1 gpurun_out/bench.log | cut -c1-900
CXG_DIGIT_KERNEL=1 timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_v1.log 2>&1; echo bench_v1=$?; tail -1 gpurun_out/bench_v1.log | cut -c1-300
timeout 400 python bench.py --steps 10 --warmup 2 --pattern '[\w]+' --synth-config 4 > gpurun_out/bench_cc.log 2>&1; echo bench_cc=$?; tail -1 gpurun_out/bench_cc.log | cut -c1-300
timeout 400 python bench.py --steps 10 --warmup 2 --pattern 'error' --synth-config 1 > gpurun_out/bench_err.log 2>&1; echo bench_err=$?; tail -1 gpurun_out/bench_err.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o ip1g -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo prof=$?
