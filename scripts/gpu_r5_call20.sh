# Round 5, call 20: config 5 — the epilogue's slots from shifts and masks decided once; the tests that go through capture rows
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; V=$R/coregex_amd/variants
O=gpurun_out/r05_c20_cfg5.txt
{
  echo "product (fast rows, lean epilogue)"; timeout 100 python scripts/time_configs.py 5 2>&1 | grep -v amdgpu.ids
  echo "variant trioold (two-word search, lean epilogue)"; CXG_LIB_PATH=$V/libcoregex_hip_trioold.so timeout 100 python scripts/time_configs.py 5 2>&1 | grep -v amdgpu.ids
  echo "product again"; timeout 100 python scripts/time_configs.py 5 2>&1 | grep -v amdgpu.ids
} > $O 2>&1
cat $O
timeout 120 python -m pytest tests/test_gpu_trio.py tests/test_gpu_golden_rows.py tests/test_gpu_parity.py -q -m gpu -x -k "trio or sub or capture or Submatch or submatch or email" 2>&1 | tail -2
