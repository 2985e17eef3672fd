# Round 6, call 21: look-around walks with ONE lookup per byte (class | kind << 8) against two; tests and fuzz of the look-around programs.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
PATS=('\b\d+\b' '\b\d+\.\d+\b' '(?m)^\d+' '\bGET\b|\bPOST\b' 'foo$|bar' '(?m)[a-z]+$' '\d+\.\d+\.\d+\.\d+\b')
for v in product prelk16 product prelk16; do
  if [ $v = product ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so; fi
  echo "== $v"; timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_c21_look_times_$v.txt | awk '{print $1, $(NF-7), $(NF-6), $(NF-5), $(NF-2)}'
done
unset CXG_LIB_PATH
timeout 900 python -m pytest tests/test_gpu_fsm.py tests/test_gpu_text_anchor.py tests/test_zz_gpu_look_wider.py tests/test_gpu_golden_rows.py tests/test_gpu_nullable.py -m gpu -q -x > gpurun_out/r06_c21_pytest.log 2>&1; echo pytest=$?; tail -3 gpurun_out/r06_c21_pytest.log | cut -c1-300
FUZZ_LOOK=1 timeout 250 python scripts/gpu_fuzz.py 682 300 > gpurun_out/r06_c21_gpu_fuzz_look.txt 2>&1; tail -1 gpurun_out/r06_c21_gpu_fuzz_look.txt | cut -c1-300
FUZZ_END=1 timeout 200 python scripts/gpu_fuzz.py 683 200 > gpurun_out/r06_c21_gpu_fuzz_end.txt 2>&1; tail -1 gpurun_out/r06_c21_gpu_fuzz_end.txt | cut -c1-300
FUZZ_TEXT=1 timeout 200 python scripts/gpu_fuzz.py 684 150 > gpurun_out/r06_c21_gpu_fuzz_text.txt 2>&1; tail -1 gpurun_out/r06_c21_gpu_fuzz_text.txt | cut -c1-300
