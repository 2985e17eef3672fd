cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
bash scripts/gpu_r6_call51.sh
timeout 200 python scripts/pair_small_sizes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c63_pair_small_sizes.txt
timeout 600 python -m pytest tests/test_gpu_teddy_pair.py -m gpu -q 2>&1 | tail -3
