cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
{
for v in s5u1 s5u2 s5u4 s5u8 pu2 pu4; do CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1; done
} > gpurun_out/r3l_modes.txt 2>&1; cat gpurun_out/r3l_modes.txt
