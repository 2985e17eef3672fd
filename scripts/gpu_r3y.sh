cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
for i in 1 2 3; do for v in ccswar cctable; do echo "$v: $(CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so timeout 300 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids | tail -1)"; done; done
timeout 600 python -m pytest -m gpu -q -x tests/test_gpu_parity.py -k "charclass or class or cc or config4 or 8gib" 2>&1 | tail -2
