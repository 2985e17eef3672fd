cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python -m pytest -m gpu -q -x tests/test_gpu_parity.py -k "use_both" > gpurun_out/r3z_both.log 2>&1; echo "both rc=$?"; tail -15 gpurun_out/r3z_both.log
timeout 300 python scripts/time_nosync.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3z_nosync.txt; cat gpurun_out/r3z_nosync.txt
