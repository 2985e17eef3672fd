# Round 6, call 85: the stretch left to the groups of 2 tiles per wave, in tile rows per CU (0 .. 8), with one counter and static first groups.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
O=gpurun_out/r06_c85_pair_tail_rows.txt; rm -f $O
for t in 1 0 2 4 8 1 0 2; do echo "tail rows $t" | tee -a $O; CXG_PAIR_TAIL_ROWS=$t timeout 120 python scripts/pair_sizes_big.py 2>&1 | grep -v amdgpu.ids | tee -a $O; done
