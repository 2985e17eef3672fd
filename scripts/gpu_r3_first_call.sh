# Round 3, first GPU call (prepared at the end of round 2, when the GPU budget was spent): the GPU tier on the final round-2 tree —
# includes tests/test_zz_gpu_look_wider.py, which only the transducer's sequential twin has seen so far — then the kernel times of
# the look-around programs that host/lookdfa.cc admits, for profiles/.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/pytest_gpu.log 2>&1; echo pytest=$?; tail -9 gpurun_out/pytest_gpu.log
timeout 600 python scripts/time_patterns.py '\d+\.\d+\.\d+\.\d+' '\d+\.\d+\.\d+\.\d+\b' '\d{4}-\d{2}-\d{2}\b' '\berror\b' '(GET|POST|PUT|DELETE)\b /[a-z/]+ HTTP' 'timeout=\d+\b ms elapsed' \
  '\b\w+=\w+;\w+=\w+\b' '\buser=\w+ ip=\w+ status=\w+\b' '[a-z]+=\d+\b; [a-z]+=\d+\b' '\b(\w+)=(\w+)\b' '(?m)^(\d+) (\w+)' > gpurun_out/time_look_dfa_patterns.txt 2>&1; cat gpurun_out/time_look_dfa_patterns.txt
# device fuzz over look-around programs (now including the UseDFA / UseBoth / UseDigitPrefilter ones that pass the proof, and captures)
FUZZ_LOOK=1 timeout 900 python scripts/gpu_fuzz.py 301 600 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r03_gpu_fuzz_look_first.txt; cat gpurun_out/r03_gpu_fuzz_look_first.txt
