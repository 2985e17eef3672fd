# Round 4 device fuzz on the final kernels: wide (`.`, negated classes: class runs, quote pairs, offset captures are in the atom list), fold, general (nullable atoms included), look
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
FUZZ_WIDE=1 timeout 400 python scripts/gpu_fuzz.py 82 300 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r04_gpu_fuzz_wide.txt; cat gpurun_out/r04_gpu_fuzz_wide.txt
timeout 300 python scripts/gpu_fuzz.py 83 300 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r04_gpu_fuzz_general.txt; cat gpurun_out/r04_gpu_fuzz_general.txt
FUZZ_FOLD=1 timeout 200 python scripts/gpu_fuzz.py 84 200 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r04_gpu_fuzz_fold.txt; cat gpurun_out/r04_gpu_fuzz_fold.txt
FUZZ_LOOK=1 timeout 200 python scripts/gpu_fuzz.py 85 200 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r04_gpu_fuzz_look.txt; cat gpurun_out/r04_gpu_fuzz_look.txt
