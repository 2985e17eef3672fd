# long device-vs-oracle fuzz campaigns (evidence: profiles/r02_gpu_fuzz_*.txt)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
for seed in ${LOOK_SEEDS:-41 42 43}; do FUZZ_LOOK=1 timeout 1500 python scripts/gpu_fuzz.py $seed ${LOOK_N:-2000} 2>&1 | grep -v amdgpu.ids | tail -6; done > gpurun_out/r02_gpu_fuzz_look.txt
cat gpurun_out/r02_gpu_fuzz_look.txt
for seed in ${GEN_SEEDS:-51 52}; do timeout 1500 python scripts/gpu_fuzz.py $seed ${GEN_N:-1500} 2>&1 | grep -v amdgpu.ids | tail -6; done > gpurun_out/r02_gpu_fuzz_general.txt
cat gpurun_out/r02_gpu_fuzz_general.txt
