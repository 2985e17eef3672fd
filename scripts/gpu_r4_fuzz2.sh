cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_text_anchor.py tests/test_zzzz_gpu_routing.py -m gpu -q -x > gpurun_out/r04_text_pytest.log 2>&1; echo pytest=$?; tail -3 gpurun_out/r04_text_pytest.log | cut -c1-300
FUZZ_TEXT=1 timeout 150 python scripts/gpu_fuzz.py 91 200 > gpurun_out/r04_gpu_fuzz_text.txt 2>&1; tail -3 gpurun_out/r04_gpu_fuzz_text.txt | cut -c1-400
FUZZ_FOLD=1 timeout 120 python scripts/gpu_fuzz.py 92 150 > gpurun_out/r04_gpu_fuzz_fold2.txt 2>&1; tail -2 gpurun_out/r04_gpu_fuzz_fold2.txt | cut -c1-400
FUZZ_WIDE=1 timeout 240 python scripts/gpu_fuzz.py 82 300 > gpurun_out/r04_gpu_fuzz_wide.txt 2>&1; tail -3 gpurun_out/r04_gpu_fuzz_wide.txt | cut -c1-400
