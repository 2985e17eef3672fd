cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; export CXG_PAIR_MIN_BYTES=0
bash scripts/gpu_r6_call51.sh
timeout 600 python -m pytest tests/test_gpu_teddy_pair.py -m gpu -q 2>&1 | tail -2 | cut -c1-300
timeout 120 python scripts/pair_sizes_big.py 2>&1 | grep -v amdgpu.ids
timeout 120 python scripts/pair_sizes_big.py 2>&1 | grep -v amdgpu.ids
