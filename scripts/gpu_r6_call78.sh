# Round 6, call 78: the persistent fields kernel with claimed units (-DCXG_PF_TICKETS=1) with and without LLVM's atomic optimizer
# (the pair kernel's first run lost 0.3 ms to the optimizer's wait behind every ticket).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
O=gpurun_out/r06_c78_fields_tickets_optimizer.txt; rm -f $O
for v in product pftk pftknone; do
  if [ $v = product ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so; fi
  echo "== $v" | tee -a $O
  timeout 200 python scripts/time_configs.py 2 2>&1 | grep -v amdgpu.ids | tee -a $O
  timeout 200 python scripts/time_configs.py 2 2>&1 | grep -v amdgpu.ids | tee -a $O
done
