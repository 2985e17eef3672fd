# Round-end evidence: GPU parity tests, the default bench line, rocprofv3 kernel statistics of the same command and of
# all five configurations.  Run on the GPU box via gpurun; summaries are copied into profiles/ by hand afterwards.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/prof gpurun_out/prof_cfgs
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest=$?; tail -3 gpurun_out/pytest_gpu.log
S=$(date +%s); timeout 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo bench=$? wall=$(( $(date +%s) - S ))s; tail -1 gpurun_out/bench.log | cut -c1-1500
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o ip1g -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo prof=$?
PYTHONPATH=$GRAFT_REPO_ROOT timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfgs -o cfgs -- python $GRAFT_REPO_ROOT/scripts/time_configs.py > $GRAFT_REPO_ROOT/gpurun_out/prof_cfgs.log 2>&1; echo prof_cfgs=$?
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/prof.log | cut -c1-400
grep cfg gpurun_out/prof_cfgs.log
python scripts/rocprof_summary.py gpurun_out/prof/ip1g_results.db gpurun_out/ip1g_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (default: 40 settle + 10 warm-up + 50 timed launches)" > /dev/null; head -12 gpurun_out/ip1g_kernel_stats.txt
python scripts/rocprof_summary.py gpurun_out/prof_cfgs/cfgs_results.db gpurun_out/all_configs_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python scripts/time_configs.py   (five BASELINE configs, 1 GiB each, 6 launches per config)" > /dev/null; head -16 gpurun_out/all_configs_kernel_stats.txt
