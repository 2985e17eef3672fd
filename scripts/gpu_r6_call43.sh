# Round 6, call 43: where the pair kernel's time goes — ablated builds (CXG_PAIR_ABL: 1 no verification, 3 no candidate list either, 4 no table lookups, 8 no look-back / row write, 15 all).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
O=gpurun_out/r06_c43_pair_abl.txt; rm -f $O
for v in product pabl1 pabl8 pabl16 pabl32; do
  if [ $v = product ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so; fi
  timeout 120 python scripts/pair_abl_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O
done
