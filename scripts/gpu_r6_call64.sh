cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
O=gpurun_out/r06_c64_pair_tail.txt; rm -f $O
for t in 0 1 2 4 8 16; do CXG_PAIR_TAIL8=$t timeout 120 python scripts/pair_sizes_big.py 2>&1 | grep -v amdgpu.ids | tee -a $O; done
CXG_NO_TEDDY_PAIR=1 CXG_PAIR_TAIL8=wave timeout 120 python scripts/pair_sizes_big.py 2>&1 | grep -v amdgpu.ids | tee -a $O
