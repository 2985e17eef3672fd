# Round 3, call B: where does the fields kernel's time go?  Ablated / re-parameterised variants (scripts/build_variant.sh), same
# command, 1 GiB of the config-2 corpus; then the kernel's tests and the GPU tier.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
P1='\d+\.\d+\.\d+\.\d+'
{
echo "== product"; timeout 200 python scripts/time_patterns.py "$P1" 2>&1 | grep -v amdgpu.ids
for v in fabl1 fabl2 fabl3 fabl4 fabl5 noreissue waves4 waves6 tpw4 tpw16 depth2; do echo "== variant $v"; CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so timeout 120 python scripts/time_patterns.py "$P1" 2>&1 | grep -v amdgpu.ids; done
echo "== product again"; timeout 200 python scripts/time_patterns.py "$P1" 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r3b_times.txt 2>&1; cat gpurun_out/r3b_times.txt
timeout 600 python -m pytest tests/test_gpu_fields.py -q > gpurun_out/r3b_fields_tests.log 2>&1; echo "fields tests rc=$?"; tail -15 gpurun_out/r3b_fields_tests.log
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r3b_gpu_tier.log 2>&1; echo "gpu tier rc=$?"; tail -8 gpurun_out/r3b_gpu_tier.log
