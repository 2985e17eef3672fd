# Round 6, call 76: the pair kernel's last stretch in groups of 4 tiles per wave (build ptpw4) against 2, by length and stretch size.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
O=gpurun_out/r06_c76_pair_tail_tpw4.txt; rm -f $O
for t in 1 2; do echo "small groups of 2 tiles, stretch $t/8" | tee -a $O; CXG_PAIR_TAIL8=$t timeout 120 python scripts/pair_sizes_big.py 2>&1 | grep -v amdgpu.ids | tee -a $O; done
export CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_ptpw4.so
for t in 1 2 4 8; do echo "small groups of 4 tiles, stretch $t/8" | tee -a $O; CXG_PAIR_TAIL8=$t timeout 120 python scripts/pair_sizes_big.py 2>&1 | grep -v amdgpu.ids | tee -a $O; done
