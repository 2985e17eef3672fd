# Round 6, call 12: where the transducer kernel's instructions go — PMC counts of the ablated builds (CXG_FSM_ABL: 1 no entry walks, 2 no lockstep walk, 4 no rows) on the README IP pattern.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
README_IP='(?:(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.){3}(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
rm -f gpurun_out/r06_c12_fsm_abl_insts.txt
for v in base abl1 abl2 abl4 abl7; do
  if [ $v = base ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so; fi
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$v && CXG_FSM_NO_DIRECT=1 timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_$v -o pmc --output-format csv -- python $R/scripts/time_patterns.py "$README_IP" > /tmp/pmc_$v.log 2>&1)
  f=$(find /tmp/pmc_$v -name "*counter_collection.csv" | head -1)
  python - "$f" "$v" <<'PY' | tee -a gpurun_out/r06_c12_fsm_abl_insts.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "k_scan" in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
tiles = (1 << 30) / 3840
for k, a in acc.items():
    print(sys.argv[2], k, "per wave-tile:", {c: round(sum(v) / len(v) / tiles, 1) for c, v in sorted(a.items()) if c.startswith("SQ_INSTS")}, "launches", len(a["SQ_WAVES"]))
PY
  grep kernel_ms /tmp/pmc_$v.log | sed -E 's/^.*(matches +[0-9]+ kernel_ms +[0-9.]+).*$/\1/'
done
