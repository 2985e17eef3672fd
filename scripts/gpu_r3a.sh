# Round 3, call A: the fields kernel (scan_fields_wave.hip) on the device for the first time.
#  1. its GPU tests   2. kernel times: product, old chain kernel (CXG_NO_FIELDS_KERNEL=1), ablated variants   3. bench line
#  4. instruction counters of the new kernel   5. the whole GPU tier
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_fields.py -x -q > gpurun_out/r3a_fields_tests.log 2>&1; echo "fields tests rc=$?"; tail -15 gpurun_out/r3a_fields_tests.log
P1='\d+\.\d+\.\d+\.\d+'; P2='\d+:\d+:\d+'; P3='\d+\.\d+'
{
echo "== product"; timeout 200 python scripts/time_patterns.py "$P1" "$P2" "$P3" 2>&1 | grep -v amdgpu.ids
echo "== CXG_NO_FIELDS_KERNEL=1 (scan_chain_wave.hip)"; CXG_NO_FIELDS_KERNEL=1 timeout 200 python scripts/time_patterns.py "$P1" "$P2" "$P3" 2>&1 | grep -v amdgpu.ids
for v in fabl1 fabl2 fabl3; do echo "== variant $v (ablation, rows wrong)"; CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so timeout 200 python scripts/time_patterns.py "$P1" 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r3a_times.txt 2>&1; cat gpurun_out/r3a_times.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; echo "bench rc=$?"; cat gpurun_out/r3a_bench.json | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_*
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_$i -o pmc --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --settle 2 --no-cpu-baseline --no-pmc > /dev/null 2>&1; echo "pmc $i rc=$?"
done
cd $R; python scripts/pmc_summary.py k_scan_fields > gpurun_out/r3a_pmc_counters.txt 2>&1; cat gpurun_out/r3a_pmc_counters.txt
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3a_gpu_tier.log 2>&1; echo "gpu tier rc=$?"; tail -5 gpurun_out/r3a_gpu_tier.log
