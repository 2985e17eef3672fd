# Round 5, call 8: config 4 — windows in flight per wave in pass 1 (1 / 2 / 4) and the two-halves extraction loop; count-only time beside
# the full time; instruction / wait counters of the product kernel.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; V=$R/coregex_amd/variants
O=gpurun_out/r05_c8_cfg4.txt
{
  echo "product"; timeout 100 python scripts/time_configs.py 4
  for v in ccd2 ccd4 ccd4h cch; do echo "variant $v"; CXG_LIB_PATH=$V/libcoregex_hip_$v.so timeout 100 python scripts/time_configs.py 4; done
  echo "product again"; timeout 100 python scripts/time_configs.py 4 5 3
} > $O 2>&1
cat $O
for v in ccd4 ccd4h; do
  CXG_LIB_PATH=$V/libcoregex_hip_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_u32_rows.py tests/test_gpu_golden_rows.py -q -m gpu -x -k "charclass or class or u32 or golden or cfg4 or config4 or pairs or quote" 2>&1 | tail -2
done
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -c4-12)
  rm -rf $R/gpurun_out/pmc_c8_$n
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_c8_$n -o pmc --output-format csv -- python $R/scripts/time_configs.py 4 > /dev/null 2>&1
done
cd $R
python - <<'PY' | tee -a gpurun_out/r05_c8_cfg4.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_c8_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "charclass" in r["Kernel_Name"]:
            out = "rows" if float(r.get("Grid_Size", 0) or 0) else "rows"
            acc[r["Dispatch_Id"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
tiles = (1 << 30) / 3840
# dispatches in order: 4 count-only launches, then 5 with rows
per = collections.defaultdict(list)
for d in sorted(acc, key=int):
    for c, v in acc[d].items():
        per[c].append(sum(v))
for c, v in sorted(per.items()):
    print(f"{c:24s} per wave-tile by dispatch:", " ".join(f"{x / tiles:9.1f}" for x in v))
PY
rm -rf gpurun_out/pmc_c8_*
