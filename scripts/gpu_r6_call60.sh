cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
O=gpurun_out/r06_c60_pair_vaux.txt; rm -f $O
for rep in 1 2; do for v in product pv2 pv1; do
  if [ $v = product ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so; fi
  timeout 120 python scripts/pair_abl_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O
done; done
