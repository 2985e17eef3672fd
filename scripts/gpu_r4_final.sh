# Last device call of round 4: smoke(), the whole GPU tier, config-2 evidence and the default bench line on the tree that is left behind.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 800 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r04_pytest_gpu.log 2>&1; echo pytest=$?; tail -14 gpurun_out/r04_pytest_gpu.log | cut -c1-300
timeout 200 python bench.py --steps 20 --warmup 3 > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo bench=$?; cut -c1-600 gpurun_out/r04_bench_final.json
{ timeout 150 python scripts/time_patterns.py '(^|\s)error' '(?:^|,)\d+' '(^|\s)(GET|POST)' 'foo|^bar' '(?i)(select|insert|update|delete)' '(?i)(error|fail|exception|panic|fatal)' 2>&1 | grep -v amdgpu.ids | tail -6 | sed 's/  */ /g'
} > gpurun_out/r04_time_final_new_programs.txt 2>&1; cat gpurun_out/r04_time_final_new_programs.txt | cut -c1-250
CFGS=2 timeout 420 bash $R/scripts/gpu_r4_evidence.sh 2>&1 | tail -6
