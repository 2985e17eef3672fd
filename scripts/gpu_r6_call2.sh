# Round 6, call 2: direct mode of the transducer kernel (byte-indexed rows) + minimised transducers — tests, A/B timings, instruction counters, fuzz.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
README_IP='(?:(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.){3}(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
timeout 600 python -m pytest tests/test_gpu_fsm.py tests/test_gpu_text_anchor.py tests/test_zz_gpu_look_wider.py tests/test_gpu_nullable.py tests/test_gpu_wide.py tests/test_zzzz_gpu_routing.py -m gpu -q > gpurun_out/r06_c2_pytest_fsm.log 2>&1; echo pytest=$?; tail -8 gpurun_out/r06_c2_pytest_fsm.log | cut -c1-300
PATS=("$README_IP" '\b\d+\b' '\b\d+\.\d+\b' '\d+\.\d+x?' '(?m)^\d+' 'a+b|b+a' 'GET|POST /[a-z]+' '(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]' '[A-Z][a-z]+ [A-Z][a-z]+')
echo "== direct mode (default)"; timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c2_fsm_times_direct.txt | cut -c1-250
echo "== CXG_FSM_NO_DIRECT=1 (class-indexed kernel, minimised tables)"; CXG_FSM_NO_DIRECT=1 timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c2_fsm_times_classidx.txt | cut -c1-250
echo "== CXG_FSM_NO_DIRECT=1 CXG_FSM_NO_MINIMISE=1 (class-indexed kernel, round-5 tables)"; CXG_FSM_NO_DIRECT=1 CXG_FSM_NO_MINIMISE=1 timeout 300 python scripts/time_patterns.py "$README_IP" '\d+\.\d+x?' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c2_fsm_times_nomin.txt | cut -c1-250
bash scripts/gpu_pmc_fsm.sh "$README_IP" r06_c2_direct 2>&1 | tail -40
CXG_FSM_NO_DIRECT=1 bash scripts/gpu_pmc_fsm.sh "$README_IP" r06_c2_classidx 2>&1 | tail -40
timeout 200 python scripts/gpu_fuzz.py 611 220 > gpurun_out/r06_c2_gpu_fuzz_general.txt 2>&1; tail -1 gpurun_out/r06_c2_gpu_fuzz_general.txt | cut -c1-300; grep -c MISMATCH gpurun_out/r06_c2_gpu_fuzz_general.txt
FUZZ_LOOK=1 timeout 150 python scripts/gpu_fuzz.py 612 150 > gpurun_out/r06_c2_gpu_fuzz_look.txt 2>&1; tail -1 gpurun_out/r06_c2_gpu_fuzz_look.txt | cut -c1-300; grep -c MISMATCH gpurun_out/r06_c2_gpu_fuzz_look.txt
FUZZ_WIDE=1 timeout 150 python scripts/gpu_fuzz.py 613 120 > gpurun_out/r06_c2_gpu_fuzz_wide.txt 2>&1; tail -1 gpurun_out/r06_c2_gpu_fuzz_wide.txt | cut -c1-300; grep -c MISMATCH gpurun_out/r06_c2_gpu_fuzz_wide.txt
