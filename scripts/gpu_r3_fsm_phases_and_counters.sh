cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
README_IP='(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
{
CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_fsmprof.so CXG_PROF=1 timeout 300 python scripts/time_patterns.py "$README_IP" 2>&1 | grep -E "CXG_PROF|kernel_ms" | tail -2 | cut -c1-400
CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_fsmprof.so CXG_PROF=1 timeout 300 python scripts/time_patterns.py '\berror\b' 2>&1 | grep -E "CXG_PROF|kernel_ms" | tail -2 | cut -c1-400
timeout 300 python scripts/time_patterns.py "$README_IP" '\berror\b' '\d+\.\d+x?' 2>&1 | grep kernel_ms
} > gpurun_out/r3p_fsm_prof.txt 2>&1; cat gpurun_out/r3p_fsm_prof.txt
bash scripts/gpu_pmc_configs.sh > gpurun_out/r3p_pmc_configs.txt 2>&1; tail -100 gpurun_out/r3p_pmc_configs.txt
