cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python scripts/gpu_dbg_pair.py 60 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c46_dbg_pair.txt | tail -30
