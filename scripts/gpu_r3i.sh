# Round 3, call D: the streaming fields kernel (persistent grid + scan server) for the first time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_fields.py -q -x > gpurun_out/r3i_fields_tests.log 2>&1; echo "fields tests rc=$?"; tail -15 gpurun_out/r3i_fields_tests.log
P1='\d+\.\d+\.\d+\.\d+'
{
echo "== product (streaming)"; timeout 200 python scripts/time_patterns.py "$P1" '\d+:\d+:\d+' 2>&1 | grep -v amdgpu.ids
echo "== CXG_FIELDS_GROUPED=1"; CXG_FIELDS_GROUPED=1 timeout 200 python scripts/time_patterns.py "$P1" 2>&1 | grep -v amdgpu.ids
for v in sabl1 sabl4 sabl5; do echo "== variant $v (streaming, ablated)"; CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so timeout 120 python scripts/time_patterns.py "$P1" 2>&1 | grep -v amdgpu.ids; done
for g in 1028 1540 2052; do echo "== CXG_FIELDS_WORKGROUPS=$g"; CXG_FIELDS_WORKGROUPS=$g timeout 120 python scripts/time_patterns.py "$P1" 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r3i_times.txt 2>&1; cat gpurun_out/r3i_times.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3i_bench.json 2> gpurun_out/r3i_bench.err; echo "bench rc=$?"; cut -c1-1200 gpurun_out/r3i_bench.json; tail -3 gpurun_out/r3i_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_*
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_$i -o pmc --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --settle 2 --no-cpu-baseline --no-pmc > /dev/null 2>&1; echo "pmc $i rc=$?"
done
cd $R; python scripts/pmc_summary.py k_scan_fields > gpurun_out/r3i_pmc_counters.txt 2>&1; cat gpurun_out/r3i_pmc_counters.txt
