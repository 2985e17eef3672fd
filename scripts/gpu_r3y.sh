cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python -m pytest -m gpu -q -x tests/test_gpu_parity.py -k "use_both" 2>&1 | tail -6 | cut -c1-300
timeout 300 python scripts/gpu_fuzz.py 95 400 2>&1 | grep -v amdgpu.ids | tail -3
