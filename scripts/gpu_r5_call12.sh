# Round 5, call 12: config 4 — the new default form (M parked, 2 windows in flight, halves, first tile staged before the look-back) against EARLY=0
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; V=$R/coregex_amd/variants
O=gpurun_out/r05_c12_cfg4.txt
{
  echo "product (EARLY=1)"; timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "variant ccne (EARLY=0)"; CXG_LIB_PATH=$V/libcoregex_hip_ccne.so timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "product, CXG_PROF=1"; CXG_PROF=1 timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tail -2
  echo "product again"; timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "other class-run patterns on the config-2 corpus"; timeout 200 python scripts/time_patterns.py '\S+' '[^,]+' '"[^"]*"' '[a-z]+' '\d+' 2>&1 | grep -v amdgpu.ids | tail -6
} > $O 2>&1
cat $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_u32_rows.py tests/test_gpu_golden_rows.py tests/test_gpu_boundary.py -q -m gpu -x 2>&1 | tail -2
