# Prepared at the end of round 4: the first device call of round 5.
#   the whole GPU tier; the golden groups added late in round 4 through the device (scripts/gpu_golden_rows.py: CPU tier only so far);
#   the general device fuzz with room (it ran into a 100 s limit in round 4's last call); kernel times of the programs served late in round 4.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 700 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r05_first_call_pytest_gpu.log 2>&1; echo pytest=$?; tail -6 gpurun_out/r05_first_call_pytest_gpu.log | cut -c1-300
timeout 200 python scripts/gpu_golden_rows.py > gpurun_out/r05_first_call_golden_rows.txt 2>&1; echo golden=$?; tail -3 gpurun_out/r05_first_call_golden_rows.txt | cut -c1-300
timeout 320 python scripts/gpu_fuzz.py 95 300 > gpurun_out/r05_first_call_gpu_fuzz_general.txt 2>&1; tail -1 gpurun_out/r05_first_call_gpu_fuzz_general.txt | cut -c1-300
{ timeout 200 python scripts/time_patterns.py '(^|\s)error' '(?:^|,)\d+' '(?i)(select|insert|update|delete)' '(?i)(error|fail|exception|panic|fatal)' '\berror\b' 2>&1 | grep -v amdgpu.ids | tail -5 | sed 's/  */ /g'; } > gpurun_out/r05_first_call_times.txt 2>&1; cat gpurun_out/r05_first_call_times.txt | cut -c1-200
