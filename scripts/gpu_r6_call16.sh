# Round 6, call 16: the whole GPU tier on the split capi; what s_memtime / s_memrealtime count; back-to-back launches of the persistent kernel.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
scripts/microbench/clocks > gpurun_out/r06_c16_clocks.txt 2>&1; cat gpurun_out/r06_c16_clocks.txt | cut -c1-220
timeout 300 python scripts/gpu_back_to_back.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c16_back_to_back.txt | cut -c1-260
CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_nontrows.so timeout 300 python scripts/gpu_back_to_back.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c16_back_to_back_default_stores.txt | cut -c1-260
timeout 1500 python -m pytest tests -m gpu -q --durations=3 > gpurun_out/r06_c16_pytest_gpu.log 2>&1; echo pytest=$?; tail -8 gpurun_out/r06_c16_pytest_gpu.log | cut -c1-300
