# PMC counters for the kernels of all five BASELINE configs (two passes; counters only, no trace domains besides kernel-trace)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/pmc_*
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/scripts/time_configs.py > /dev/null 2>&1; echo "pmc $i rc=$?"
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py k_ 2>&1 | grep -v fill_synth | tail -120
