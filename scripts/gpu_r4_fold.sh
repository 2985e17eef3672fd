cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 240 python -m pytest tests/test_gpu_wrapped.py tests/test_zzz_gpu_fold.py tests/test_zzzz_gpu_routing.py -m gpu -q -x --durations=5 > gpurun_out/r04_fold_pytest.log 2>&1; echo pytest=$?; tail -14 gpurun_out/r04_fold_pytest.log | cut -c1-600
{ timeout 250 python scripts/time_patterns.py '(?i)(error|fail|exception|panic|fatal)' '(?i)(googlebot|bingbot|yandexbot)' '(?i)(jan|feb|mar|apr|may|jun|jul|aug|sep|oct|nov|dec)' '(?i)\b(error|fail|exception|panic|fatal)\b' '(?i)error' 2>&1 | grep -v amdgpu.ids | tail -5
  echo "same, CXG_NO_WRAPPED_LITERALS=1:"
  CXG_NO_WRAPPED_LITERALS=1 timeout 250 python scripts/time_patterns.py '(?i)(error|fail|exception|panic|fatal)' '(?i)(googlebot|bingbot|yandexbot)' '(?i)(jan|feb|mar|apr|may|jun|jul|aug|sep|oct|nov|dec)' 2>&1 | grep -v amdgpu.ids | tail -3
} > gpurun_out/r04_time_fold.txt 2>&1; cat gpurun_out/r04_time_fold.txt | cut -c1-300
