# Round 5, call 14: config 4 — is the chain of groups the bound?  8 / 4 / 2 tiles per wave = 120 / 60 / 30 KiB groups (8.7 k / 17.5 k / 35 k groups per GiB)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; V=$R/coregex_amd/variants
O=gpurun_out/r05_c14_cfg4_group_size.txt
{
  echo "product (4 tiles per wave, 6 workgroups per CU)"; timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "variant cct8 (8 tiles per wave, 4 workgroups per CU)"; CXG_LIB_PATH=$V/libcoregex_hip_cct8.so timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "variant cct2 (2 tiles per wave, 8 workgroups per CU)"; CXG_LIB_PATH=$V/libcoregex_hip_cct2.so timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "variant cct8, CXG_PROF=1"; CXG_PROF=1 CXG_LIB_PATH=$V/libcoregex_hip_cct8.so timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tail -2
  echo "product again"; timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
} > $O 2>&1
cat $O
