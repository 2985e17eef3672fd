# Round 3, call C: look-back width and haystack-to-workgroup mapping (ablations; variants built by scripts/build_variant.sh)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
P1='\d+\.\d+\.\d+\.\d+'
{
echo "== product (wide look-back)"; timeout 200 python scripts/time_patterns.py "$P1" 2>&1 | grep -v amdgpu.ids
for v in narrowlb fabl3w fabl4w fabl5 fabl5n fabl5m fabl5mn fabl3m map1 depth2; do echo "== variant $v"; CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so timeout 120 python scripts/time_patterns.py "$P1" 2>&1 | grep -v amdgpu.ids; done
echo "== product again"; timeout 200 python scripts/time_patterns.py "$P1" 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r3c_times.txt 2>&1; cat gpurun_out/r3c_times.txt
timeout 600 python -m pytest tests/test_gpu_fields.py -q -x > gpurun_out/r3c_fields_tests.log 2>&1; echo "fields tests rc=$?"; tail -3 gpurun_out/r3c_fields_tests.log
