cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_fields.py -q -x > gpurun_out/r3n_fields_tests.log 2>&1; echo "fields tests rc=$?"; tail -5 gpurun_out/r3n_fields_tests.log
{
CXG_PROF=1 timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | grep PROF | tail -2
timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
CXG_FIELDS_GROUPED=1 timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
for v in noprio ring512 sg4 sg2 sabl4; do CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1; done
} > gpurun_out/r3n_modes.txt 2>&1; cat gpurun_out/r3n_modes.txt
