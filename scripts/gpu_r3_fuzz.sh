# round-3 device-vs-oracle fuzz campaigns on the final tree (evidence: profiles/r03_gpu_fuzz_*.txt)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
FUZZ_FEW=1 timeout 900 python scripts/gpu_fuzz.py 61 ${FEW_N:-250} 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r03_gpu_fuzz_few_symbols.txt; cat gpurun_out/r03_gpu_fuzz_few_symbols.txt
timeout 900 python scripts/gpu_fuzz.py 62 ${GEN_N:-800} 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r03_gpu_fuzz_general.txt; cat gpurun_out/r03_gpu_fuzz_general.txt
FUZZ_LOOK=1 timeout 900 python scripts/gpu_fuzz.py 63 ${LOOK_N:-800} 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r03_gpu_fuzz_look.txt; cat gpurun_out/r03_gpu_fuzz_look.txt
