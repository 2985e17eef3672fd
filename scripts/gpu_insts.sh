# instruction counts of the headline kernel (one PMC pass) + bench time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/pmc_i
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc $BENCH_ARGS > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/pmc_i/pmc_counter_collection.csv")):
    if "k_scan" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
tiles = (1 << 30) / 3840
print("per wave-tile:", {k: round(sum(v) / len(v) / tiles, 1) for k, v in acc.items() if k.startswith("SQ_INSTS")}, "GRBM cycles/8:", round(sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8))
PY
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc $BENCH_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('IP', d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
