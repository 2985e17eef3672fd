# Round 6, call 44: the pair kernel with 1 / 2 / 4 / 8 group counters (strict ticket order against spread atomics).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
O=gpurun_out/r06_c44_pair_nctr.txt; rm -f $O
for n in 8 4 2 1; do echo "counters $n" | tee -a $O; CXG_PAIR_NCTR=$n timeout 120 python scripts/pair_abl_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O; done
export CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_pabl8.so
for n in 8 1; do echo "no look-back, counters $n" | tee -a $O; CXG_PAIR_NCTR=$n timeout 120 python scripts/pair_abl_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O; done
