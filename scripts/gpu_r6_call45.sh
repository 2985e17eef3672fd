# Round 6, call 45: the pair kernel with bases resolved two units late.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
O=gpurun_out/r06_c45_pair.txt; rm -f $O
timeout 200 python scripts/teddy_pair_ab.py cfg3 four fold 2>&1 | grep -v amdgpu.ids | tee -a $O
CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_pabl8.so timeout 120 python scripts/pair_abl_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrapped.py tests/test_zzz_gpu_fold.py -m gpu -q -k "teddy or edge_cases or reference_corpus or wrapped or fold or literal" > gpurun_out/r06_c45_pytest_teddy.log 2>&1; echo pytest=$?; tail -3 gpurun_out/r06_c45_pytest_teddy.log | cut -c1-300
