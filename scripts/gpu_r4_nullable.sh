cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_nullable.py tests/test_gpu_wide.py tests/test_zzz_gpu_fold.py -m gpu -q -x > gpurun_out/r04_nullable_pytest.log 2>&1; echo pytest=$?; tail -8 gpurun_out/r04_nullable_pytest.log | cut -c1-600
{ timeout 200 python scripts/time_patterns.py '\S+' '[^,]+' '[^"]+' 2>&1 | grep -v amdgpu.ids | tail -5; } > gpurun_out/r04_time_class_runs.txt; cat gpurun_out/r04_time_class_runs.txt
