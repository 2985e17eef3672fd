# Round 6, call 1: the transducer kernel's new row derivation (shallow machines: rows from the event bits) — its GPU tests, timings, phases, device fuzz.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
README_IP='(?:(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.){3}(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
timeout 600 python -m pytest tests/test_gpu_fsm.py tests/test_gpu_text_anchor.py tests/test_zz_gpu_look_wider.py tests/test_gpu_nullable.py tests/test_gpu_wide.py -m gpu -x -q > gpurun_out/r06_c1_pytest_fsm.log 2>&1; echo pytest=$?; tail -5 gpurun_out/r06_c1_pytest_fsm.log | cut -c1-300
timeout 300 python scripts/time_patterns.py "$README_IP" '\b\d+\b' '\b\d+\.\d+\b' '\d+\.\d+x?' '(?m)^\d+' '\berror\b' 'a+b|b+a' 'https?://[^ ]+' > gpurun_out/r06_c1_fsm_times.txt 2>&1; cat gpurun_out/r06_c1_fsm_times.txt | cut -c1-250
CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_fsmprof.so CXG_PROF=1 timeout 200 python scripts/time_patterns.py "$README_IP" '\b\d+\b' 2>&1 | grep -E "CXG_PROF|kernel_ms" | cut -c1-300 > gpurun_out/r06_c1_fsm_phases.txt; cat gpurun_out/r06_c1_fsm_phases.txt
timeout 200 python scripts/gpu_fuzz.py 601 200 > gpurun_out/r06_c1_gpu_fuzz_general.txt 2>&1; tail -1 gpurun_out/r06_c1_gpu_fuzz_general.txt | cut -c1-300; grep -c MISMATCH gpurun_out/r06_c1_gpu_fuzz_general.txt
FUZZ_LOOK=1 timeout 150 python scripts/gpu_fuzz.py 602 150 > gpurun_out/r06_c1_gpu_fuzz_look.txt 2>&1; tail -1 gpurun_out/r06_c1_gpu_fuzz_look.txt | cut -c1-300; grep -c MISMATCH gpurun_out/r06_c1_gpu_fuzz_look.txt
