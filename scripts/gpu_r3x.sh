cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
for i in 1 2 3; do
for v in tfold tnofold; do echo "$v: $(CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so timeout 300 python scripts/time_configs.py 3 2>&1 | grep -v amdgpu.ids | tail -1)"; done
done
CXG_TICKETS=1 timeout 600 python -m pytest tests/test_gpu_fields.py -x -q 2>&1 | tail -1
