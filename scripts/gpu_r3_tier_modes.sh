# the whole GPU tier under the non-default status paths (no -x: tests that assert mode-specific facts may fail; look at which)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
{
for v in "CXG_NO_EPOCH=1" "CXG_TICKETS=1"; do
  echo "== $v: pytest tests -m gpu"; env $v timeout 1500 python -m pytest tests -m gpu -q --deselect "tests/test_gpu_parity.py::test_8gib_count_and_checksum_vs_multithreaded_oracle" 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-250
done
} > gpurun_out/r03_gpu_tier_modes.txt 2>&1; cat gpurun_out/r03_gpu_tier_modes.txt
