cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 800 python scripts/gpu_stress_pair.py 20000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c48_stress_pair.txt | tail -30
