# PMC passes over the transducer kernel on `1.1.1...` (64 MiB): instruction counts and wait buckets.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_*
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_1 -o pmc --output-format csv -- python $R/scripts/time_nosync_sizes.py '\d+\.\d+x?' 64 > $R/gpurun_out/pmc1.log 2>&1; echo pmc1=$?
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_2 -o pmc --output-format csv -- python $R/scripts/time_nosync_sizes.py '\d+\.\d+x?' 64 > $R/gpurun_out/pmc2.log 2>&1; echo pmc2=$?
timeout 300 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_FLAT -d $R/gpurun_out/pmc_3 -o pmc --output-format csv -- python $R/scripts/time_nosync_sizes.py '\d+\.\d+x?' 64 > $R/gpurun_out/pmc3.log 2>&1; echo pmc3=$?
cd $R
for d in pmc_1 pmc_2 pmc_3; do f=$(find gpurun_out/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/$d/pmc_counter_collection.csv; done
python scripts/pmc_summary.py k_scan_fsm > gpurun_out/pmc_fsm_nosync.txt 2>&1; cat gpurun_out/pmc_fsm_nosync.txt
tail -2 gpurun_out/pmc1.log; tail -2 gpurun_out/pmc3.log
