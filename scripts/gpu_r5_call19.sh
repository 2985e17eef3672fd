# Round 5, call 19: config 5 — capture rows from the K highest bits of one word (trio_rows) against the two-word search; the trio tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; V=$R/coregex_amd/variants
O=gpurun_out/r05_c19_cfg5.txt
{
  echo "product (fast rows)"; timeout 100 python scripts/time_configs.py 5 2>&1 | grep -v amdgpu.ids
  echo "variant trioold"; CXG_LIB_PATH=$V/libcoregex_hip_trioold.so timeout 100 python scripts/time_configs.py 5 2>&1 | grep -v amdgpu.ids
  echo "product again"; timeout 100 python scripts/time_configs.py 5 2>&1 | grep -v amdgpu.ids
} > $O 2>&1
cat $O
timeout 150 python -m pytest tests/test_gpu_trio.py -q -m gpu -x 2>&1 | tail -2
