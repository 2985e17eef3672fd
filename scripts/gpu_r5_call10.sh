# Round 5, call 10: config 4 — membership words parked (6 workgroups per CU), lean extraction and row loops
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; V=$R/coregex_amd/variants
O=gpurun_out/r05_c10_cfg4.txt
{
  echo "product"; timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  for v in cch2 ccpm ccpm1 ccpm5 ccpmnh; do echo "variant $v"; CXG_LIB_PATH=$V/libcoregex_hip_$v.so timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids; done
  echo "variant ccpm, CXG_PROF=1"; CXG_PROF=1 CXG_LIB_PATH=$V/libcoregex_hip_ccpm.so timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tail -3
  echo "product again"; timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
} > $O 2>&1
cat $O
for v in ccpm cch2; do
  CXG_LIB_PATH=$V/libcoregex_hip_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_u32_rows.py tests/test_gpu_golden_rows.py -q -m gpu -x -k "charclass or class or u32 or golden or cfg4 or config4 or pairs or quote" 2>&1 | tail -2
done
