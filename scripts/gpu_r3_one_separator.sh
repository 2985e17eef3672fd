cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
# capture rows of programs with one separator for every link (k_scan_trio_wave<K, true>), and the epilogue that gives every lane one pair of a row
timeout 270 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu.log 2>&1; echo pytest=$?; grep -E "^FAILED|passed|failed|Error" gpurun_out/r03_pytest_gpu.log | cut -c1-300 | head -20
{ echo "scripts/time_patterns.py on 1 GiB of the config-2 corpus, FindAllSubmatchIndex (round 3, k_scan_trio_wave<K, EQ>)";
  timeout 100 python scripts/time_patterns.py '(\d+)\.(\d+)\.(\d+)\.(\d+)' '(\d+)\.(\d+)\.(\d+)' '(\w+)=(\w+)' '(\w+)@(\w+)\.(\w+)' 2>&1 | tail -6;
  echo "CXG_NO_TRIO_KERNEL=1:";
  CXG_NO_TRIO_KERNEL=1 timeout 60 python scripts/time_patterns.py '(\d+)\.(\d+)\.(\d+)\.(\d+)' 2>&1 | tail -3; } > gpurun_out/r03_time_one_separator.txt
cat gpurun_out/r03_time_one_separator.txt
