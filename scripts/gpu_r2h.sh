# instruction counters of the transducer kernel (production build) for a few patterns
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for PAT in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d /tmp/pmc_$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/scripts/time_patterns.py "$PAT" > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  python - "$f" "$PAT" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_scan" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
tiles = (1 << 30) / 3840
print(sys.argv[2][:40], {k: round(sum(v) / len(v) / tiles, 1) for k, v in sorted(acc.items())})
PY
  grep kernel_ms /tmp/pmc_$i.log | sed -E 's/^.*(kernel_ms +[0-9.]+).*$/\1/'
done
