cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 300 python scripts/time_nosync.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3x_nosync.txt; cat gpurun_out/r3x_nosync.txt
timeout 600 python -m pytest tests/test_gpu_fields.py -q -x > gpurun_out/r3x_fields_tests.log 2>&1; echo "fields tests rc=$?"; tail -5 gpurun_out/r3x_fields_tests.log
timeout 1500 python -m pytest -m gpu -q -x tests/test_gpu_fsm.py tests/test_zz_gpu_look_wider.py > gpurun_out/r3x_gpu_fsm.log 2>&1; echo "gpu fsm rc=$?"; tail -8 gpurun_out/r3x_gpu_fsm.log
timeout 300 python scripts/time_configs.py 2>&1 | grep -v amdgpu.ids | tail -12
