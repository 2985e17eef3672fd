cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
{
for v in "CXG_FSM_DEEP=1" "CXG_NO_FIELDS_KERNEL=1"; do
  echo "== $v: pytest tests -m gpu"; env $v timeout 1500 python -m pytest tests -m gpu -q --deselect "tests/test_gpu_parity.py::test_8gib_count_and_checksum_vs_multithreaded_oracle" 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-250
done
echo "== fuzz general"; timeout 900 python scripts/gpu_fuzz.py 91 1500 2>&1 | grep -v amdgpu | tail -3
echo "== fuzz general, CXG_NO_FIELDS_KERNEL=1"; CXG_NO_FIELDS_KERNEL=1 timeout 900 python scripts/gpu_fuzz.py 92 800 2>&1 | grep -v amdgpu | tail -3
echo "== fuzz few symbols"; FUZZ_FEW=1 timeout 900 python scripts/gpu_fuzz.py 93 300 2>&1 | grep -v amdgpu | tail -3
} > gpurun_out/r03_gpu_tier_modes2.txt 2>&1; cat gpurun_out/r03_gpu_tier_modes2.txt
