cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
{
CXG_PROF=1 timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -4
for v in sabl1 sabl4 sabl5 sx2 sx4 sx6 sx14 p2 p6; do CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1; done
} > gpurun_out/r3j_modes.txt 2>&1; cat gpurun_out/r3j_modes.txt
