cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
O=gpurun_out/r06_c83_pair_sizes_final.txt; rm -f $O
timeout 200 python scripts/pair_small_sizes.py 2>&1 | grep -v amdgpu.ids | tee -a $O
CXG_NO_TEDDY_PAIR=1 timeout 200 python scripts/pair_small_sizes.py 2>&1 | grep -v amdgpu.ids | tee -a $O
timeout 120 python scripts/pair_sizes_big.py 2>&1 | grep -v amdgpu.ids | tee -a $O
CXG_NO_TEDDY_PAIR=1 CXG_PAIR_TAPER=wave timeout 120 python scripts/pair_sizes_big.py 2>&1 | grep -v amdgpu.ids | tee -a $O
