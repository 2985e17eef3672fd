# First device call of round 4: everything round 3 added after its GPU minutes ran out, then the standing evidence.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/r04_pytest_gpu.log 2>&1; echo pytest=$?; grep -E "^FAILED|passed|failed|Error" gpurun_out/r04_pytest_gpu.log | cut -c1-300 | head -20
# (?i) programs and unanchored UseBoundedBacktracker programs never ran on a device in round 3 (tests/test_zzz_gpu_fold.py is their tier): parity first, then the fuzz over them, then their kernel times
FUZZ_FOLD=1 timeout 200 python scripts/gpu_fuzz.py 81 300 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r04_gpu_fuzz_fold.txt; cat gpurun_out/r04_gpu_fuzz_fold.txt
FUZZ_WIDE=1 timeout 200 python scripts/gpu_fuzz.py 82 300 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r04_gpu_fuzz_wide.txt; cat gpurun_out/r04_gpu_fuzz_wide.txt
{ echo "scripts/time_patterns.py on 1 GiB of the config-2 corpus (round 4, first call)";
  timeout 200 python scripts/time_patterns.py '(?i)error' '(?i)(?:login|auth|sign.?in|session)' '(?i)(error|fail|exception|panic|fatal)' '"[^"]*"' 'GET .* HTTP' '\[[^\]]+\]' 'user=(\S+)' '<[^>]+>' 'https?://[^\s]+' '(\d+)\.(\d+)\.(\d+)\.(\d+)' '\S+' '[0-9a-f]{32}' '[^,]+' 2>&1 | grep -v amdgpu.ids | tail -14; } > gpurun_out/r04_time_new_programs.txt; cat gpurun_out/r04_time_new_programs.txt
timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r04_bench_default.json 2>/dev/null; cat gpurun_out/r04_bench_default.json
