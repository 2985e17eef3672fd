cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python -m pytest -m gpu -q -x tests/test_gpu_parity.py -k "use_both" > gpurun_out/r3u_both.log 2>&1; echo "both rc=$?"; tail -6 gpurun_out/r3u_both.log
FUZZ_FEW=1 timeout 900 python scripts/gpu_fuzz.py 64 200 2>&1 | grep -v amdgpu.ids | tail -4
FUZZ_FEW=1 timeout 900 python scripts/gpu_fuzz.py 65 250 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python scripts/gpu_fuzz.py 66 500 2>&1 | grep -v amdgpu.ids | tail -4
