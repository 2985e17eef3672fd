# Round 4: the persistent fields kernel (k_scan_fields_pers) — parity tier of the fields programs, then A/B against the grouped kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_fields.py -m gpu -q -x > gpurun_out/r04_pers_pytest.log 2>&1; echo pytest=$?; tail -5 gpurun_out/r04_pers_pytest.log | cut -c1-400
{ for occ in 6 5 4; do echo "7 tiles per unit, CXG_PF_OCC=$occ"; CXG_PF_OCC=$occ timeout 120 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1; done
  echo "verbose (statistics)"; CXG_VERBOSE=1 timeout 120 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -2
  echo "CXG_NO_PERSIST=1"; CXG_NO_PERSIST=1 timeout 120 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
  for v in pf1 t8 t9 t5; do echo "variant $v"; CXG_LIB_PATH=$GRAFT_REPO_ROOT/coregex_amd/variants/libcoregex_hip_$v.so timeout 120 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1; done
} > gpurun_out/r04_pers_ab.txt 2>&1; cat gpurun_out/r04_pers_ab.txt
