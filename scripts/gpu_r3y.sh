cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
{
timeout 300 python scripts/time_nosync_sizes.py 2>&1 | grep -v amdgpu.ids
CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_fsmprof.so CXG_PROF=1 timeout 300 python scripts/time_nosync_sizes.py "\\d+\\.\\d+x?" 64 2>&1 | grep -v amdgpu.ids | grep -E "CXG_PROF| (64) MiB" | tail -3 | cut -c1-600
} > gpurun_out/r3y_nosync_prof.txt 2>&1; cat gpurun_out/r3y_nosync_prof.txt
timeout 1500 python -m pytest -m gpu -q -x tests/test_gpu_fsm.py tests/test_zz_gpu_look_wider.py > gpurun_out/r3x_gpu_fsm.log 2>&1; echo "gpu fsm rc=$?"; tail -4 gpurun_out/r3x_gpu_fsm.log
