cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
{
for v in sl8 sl32 sl100; do for m in "" "CXG_FIELDS_GROUPED=1"; do echo "== $v $m"; env $m CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1; done; done
echo "== product grouped"; CXG_FIELDS_GROUPED=1 timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
} > gpurun_out/r3s_modes.txt 2>&1; cat gpurun_out/r3s_modes.txt
