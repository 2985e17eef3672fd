# instruction counts of ablated builds of the transducer kernel (scripts: make EXTRA=-DCXG_FSM_ABL=n OUT=../libcxg_ablN.so)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
PAT="$1"
cd /tmp && export TMPDIR=/tmp
for v in base abl2 abl4; do
  if [ $v = base ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$GRAFT_REPO_ROOT/coregex_amd/libcxg_$v.so; fi
  rm -rf /tmp/pmc_$v
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_$v -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/scripts/time_patterns.py "$PAT" > /tmp/pmc_$v.log 2>&1
  f=$(find /tmp/pmc_$v -name "*counter_collection.csv" | head -1)
  python - "$f" $v <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_scan_fsm" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
tiles = (1 << 30) / 3840
print(sys.argv[2], {k: round(sum(v) / len(v) / tiles, 1) for k, v in sorted(acc.items())})
PY
  grep kernel_ms /tmp/pmc_$v.log | sed -E 's/^.*(kernel_ms +[0-9.]+).*$/\1/'
done
