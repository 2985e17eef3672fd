# Round 5, call 11: what the char-class kernel waits for — SQ cycle counters, LDS and TA FIFOs, L2 write path (variant ccpm and product)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; V=$R/coregex_amd/variants
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_WRREQ_STALL_sum TCC_EA_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_WRITEBACK_sum" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  for lib in product ccpm; do
    rm -rf /tmp/pmc_${lib}_$i
    if [ $lib = product ]; then L=""; else L=$V/libcoregex_hip_$lib.so; fi
    CXG_LIB_PATH=$L timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_${lib}_$i -o pmc --output-format csv -- python $R/scripts/time_configs.py 4 > /tmp/pmc_${lib}_$i.log 2>&1 || { echo "pass $i $lib failed"; tail -3 /tmp/pmc_${lib}_$i.log; }
  done
done
cd $R
python - <<'PY' | tee gpurun_out/r05_c11_cfg4_counters.txt
import csv, glob, collections
tiles = (1 << 30) / 3840
for lib in ("product", "ccpm"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"/tmp/pmc_{lib}_*/**/pmc_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "charclass" in r["Kernel_Name"]:
                acc[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    print(f"== {lib}: per wave-tile (279 620 tiles), count-only launches | launches with rows")
    for c in sorted(acc):
        v = [acc[c][d] for d in sorted(acc[c])]
        cnt, rows = v[:4], v[4:]
        print(f"{c:42s} {sum(cnt) / max(1, len(cnt)) / tiles:12.2f} | {sum(rows) / max(1, len(rows)) / tiles:12.2f}")
PY
