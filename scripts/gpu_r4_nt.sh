# Round 4: nontemporal row stores in every kernel — parity tier of the wave kernels, then the five configs with and without
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_gpu_fields.py tests/test_gpu_trio.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r04_nt_pytest.log 2>&1; echo pytest=$?; tail -3 gpurun_out/r04_nt_pytest.log | cut -c1-400
{ echo "product: nontemporal rows, persistent fields kernel"; timeout 200 python scripts/time_configs.py 2>&1 | grep -v amdgpu.ids | tail -5
  echo "CXG_NO_PERSIST=1 (grouped fields kernel, nontemporal rows)"; CXG_NO_PERSIST=1 timeout 200 python scripts/time_configs.py 2 2>&1 | grep -v amdgpu.ids | tail -1
  echo "variant nont (-DCXG_NO_NT_ROWS: rows through the L2's write-back path, as in round 3)"; CXG_LIB_PATH=$GRAFT_REPO_ROOT/coregex_amd/variants/libcoregex_hip_nont.so timeout 200 python scripts/time_configs.py 2>&1 | grep -v amdgpu.ids | tail -5
  echo "README IP and word boundary (k_scan_fsm)"; timeout 200 python scripts/time_patterns.py '(?:(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.){3}(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)' '\berror\b' '\S+' 2>&1 | grep -v amdgpu.ids | tail -3
} > gpurun_out/r04_nt_configs.txt 2>&1; cat gpurun_out/r04_nt_configs.txt
