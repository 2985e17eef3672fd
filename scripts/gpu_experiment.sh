# Timing experiments for the digit kernels (debug switches + PMC counters).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "CXG_DEBUG=0" "CXG_DEBUG=1" "CXG_DEBUG=4" "CXG_DEBUG=8" "CXG_DEBUG=12"; do
  echo "== gen3 $v"; env $v timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'])"
done
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$n -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "pmc $n rc=$?"
done
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_*/ | head -30
