cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
{ for occ in 6 5 4; do echo "product (7 tiles per unit), CXG_PF_OCC=$occ"; CXG_PF_OCC=$occ timeout 120 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1; done
  for v in t9 t5 t11 noprio pf1; do echo "variant $v"; CXG_LIB_PATH=$GRAFT_REPO_ROOT/coregex_amd/variants/libcoregex_hip_$v.so timeout 120 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1; done
  echo "variant t9 occ 5"; CXG_PF_OCC=5 CXG_LIB_PATH=$GRAFT_REPO_ROOT/coregex_amd/variants/libcoregex_hip_t9.so timeout 120 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
  echo "verbose"; CXG_VERBOSE=1 timeout 120 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | grep -v XCD | tail -3
} > gpurun_out/r04_pers_tune_nt.txt 2>&1; cat gpurun_out/r04_pers_tune_nt.txt
