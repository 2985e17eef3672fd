cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
FUZZ_WIDE=1 timeout 55 python scripts/gpu_fuzz.py 71 ${WIDE_N:-100} 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r03_gpu_fuzz_wide.txt; cat gpurun_out/r03_gpu_fuzz_wide.txt
{ echo "scripts/time_patterns.py on 1 GiB of the config-2 corpus (round 3): patterns with '.' and negated classes";
  timeout 32 python scripts/time_patterns.py '"[^"]*"' 'GET .* HTTP' '\[[^\]]+\]' 'user=(\S+)' 2>&1 | grep -v amdgpu.ids | tail -5; } > gpurun_out/r03_time_wide.txt; cat gpurun_out/r03_time_wide.txt
