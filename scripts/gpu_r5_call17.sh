# Round 5, call 17: config 4 — static groups without the claim barrier; first tile staged in front of the look-back again (EARLY=1), now that pass 1 is short
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; V=$R/coregex_amd/variants
O=gpurun_out/r05_c17_cfg4.txt
{
  echo "product"; timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "variant ccearly"; CXG_LIB_PATH=$V/libcoregex_hip_ccearly.so timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "product, CXG_PROF=1"; CXG_PROF=1 timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tail -2
  echo "variant ccearly, CXG_PROF=1"; CXG_PROF=1 CXG_LIB_PATH=$V/libcoregex_hip_ccearly.so timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tail -2
  echo "product again"; timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "tickets"; CXG_TICKETS=1 timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
} > $O 2>&1
cat $O
