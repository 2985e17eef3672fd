# Round 5, call 15: config 4 — pass 2 without exec-masked branches (four streams per lane and iteration, dump slot), row loop without per-lane cases
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
O=gpurun_out/r05_c15_cfg4.txt
{
  echo "product"; timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "product, CXG_PROF=1"; CXG_PROF=1 timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tail -2
  echo "product again"; timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "other class-run patterns on the config-2 corpus"; timeout 200 python scripts/time_patterns.py '\S+' '[^,]+' '"[^"]*"' '[a-z]+' '\d+' 2>&1 | grep -v amdgpu.ids | tail -6
} > $O 2>&1
cat $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_u32_rows.py tests/test_gpu_golden_rows.py -q -m gpu -x 2>&1 | tail -3
