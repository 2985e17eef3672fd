cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_trio.py -q -x > gpurun_out/r3y_trio.log 2>&1; echo "trio rc=$?"; tail -4 gpurun_out/r3y_trio.log | cut -c1-300
for i in 1 2; do timeout 300 python scripts/time_configs.py 5 2>&1 | grep -v amdgpu.ids | tail -1; done
