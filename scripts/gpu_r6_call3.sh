# Round 6, call 3: reverse walks with prefetched groups, direct mode capped at 12 KiB — whole GPU tier, timings, fuzz.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
README_IP='(?:(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.){3}(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
PATS=("$README_IP" '\b\d+\b' '\b\d+\.\d+\b' '\d+\.\d+x?' '(?m)^\d+' 'a+b|b+a' '(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]')
timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c3_fsm_times.txt | cut -c1-250
timeout 900 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r06_c3_pytest_gpu.log 2>&1; echo pytest=$?; tail -12 gpurun_out/r06_c3_pytest_gpu.log | cut -c1-300
timeout 200 python scripts/gpu_fuzz.py 621 220 > gpurun_out/r06_c3_gpu_fuzz_general.txt 2>&1; tail -1 gpurun_out/r06_c3_gpu_fuzz_general.txt | cut -c1-300; grep -c MISMATCH gpurun_out/r06_c3_gpu_fuzz_general.txt
FUZZ_LOOK=1 timeout 150 python scripts/gpu_fuzz.py 622 150 > gpurun_out/r06_c3_gpu_fuzz_look.txt 2>&1; tail -1 gpurun_out/r06_c3_gpu_fuzz_look.txt | cut -c1-300; grep -c MISMATCH gpurun_out/r06_c3_gpu_fuzz_look.txt
