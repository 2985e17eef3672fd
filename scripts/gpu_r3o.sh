cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
{
echo "== grouped (default)"; timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "== grouped, CXG_DEBUG=2 (no look-back; rows misplaced)"; CXG_DEBUG=2 timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "== streaming (CXG_FIELDS_STREAM=1)"; CXG_FIELDS_STREAM=1 timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
} > gpurun_out/r3o_modes.txt 2>&1; cat gpurun_out/r3o_modes.txt
cd /tmp && export TMPDIR=/tmp
for mode in grouped stream; do
rm -rf $R/gpurun_out/pmc_*
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  if [ $mode = stream ]; then export CXG_FIELDS_STREAM=1; else unset CXG_FIELDS_STREAM; fi
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_$i -o pmc --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --settle 2 --no-cpu-baseline --no-pmc > /dev/null 2>&1; echo "pmc $mode $i rc=$?"
done
(cd $R; python scripts/pmc_summary.py k_scan_fields > gpurun_out/r3o_pmc_$mode.txt 2>&1; cat gpurun_out/r3o_pmc_$mode.txt)
done
