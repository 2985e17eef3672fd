# PMC passes over the transducer kernel: instruction counts and wait buckets.  usage: bash scripts/gpu_pmc_fsm.sh 'PATTERN' tag
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; PAT="$1"; TAG="$2"
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_*
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_1 -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/scripts/time_patterns.py "$PAT" > $GRAFT_REPO_ROOT/gpurun_out/pmc1.log 2>&1; echo pmc1=$?
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -d $GRAFT_REPO_ROOT/gpurun_out/pmc_2 -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/scripts/time_patterns.py "$PAT" > $GRAFT_REPO_ROOT/gpurun_out/pmc2.log 2>&1; echo pmc2=$?
cd $GRAFT_REPO_ROOT
find gpurun_out/pmc_1 gpurun_out/pmc_2 -name "*counter_collection.csv" | head
for d in pmc_1 pmc_2; do f=$(find gpurun_out/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/$d/pmc_counter_collection.csv; done
python scripts/pmc_summary.py k_scan_fsm > gpurun_out/pmc_fsm_$TAG.txt 2>&1; cat gpurun_out/pmc_fsm_$TAG.txt
tail -2 gpurun_out/pmc1.log
