# Round 5, call 16: config 4 — class description as kernel arguments; 60 KiB against 120 KiB groups again, now that pass 1 is not the bound
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; V=$R/coregex_amd/variants
O=gpurun_out/r05_c16_cfg4.txt
{
  echo "product"; timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "product, CXG_PROF=1"; CXG_PROF=1 timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tail -2
  echo "variant cct8 (8 tiles per wave, 4 workgroups per CU)"; CXG_LIB_PATH=$V/libcoregex_hip_cct8.so timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "variant cct8, CXG_PROF=1"; CXG_PROF=1 CXG_LIB_PATH=$V/libcoregex_hip_cct8.so timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tail -2
  echo "product again"; timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
} > $O 2>&1
cat $O
