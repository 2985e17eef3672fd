# Round-4 evidence on the current tree, per BASELINE config N = 1..5 (1 GiB, the bench.py command):
#   gpurun_out/r04_cfgN_bench.json          bench.py line (roofline + cpu_baseline)
#   gpurun_out/r04_cfgN_kernel_stats.txt    rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/r04_cfgN_pmc_traffic.json    FETCH_SIZE / WRITE_SIZE in separate --pmc passes (kernel trace only beside them)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for N in ${CFGS:-1 2 3 4 5}; do
  timeout 400 python $R/bench.py --config $N --steps 20 --warmup 5 > $R/gpurun_out/r04_cfg${N}_bench.json 2> $R/gpurun_out/r04_cfg${N}_bench.err; echo "cfg $N bench rc=$?"
  rm -rf /tmp/prof_$N
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$N -o cfg$N -- python $R/bench.py --config $N --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > /tmp/prof_$N.log 2>&1; echo "cfg $N stats rc=$?"
  db=$(find /tmp/prof_$N -name "*.db" | head -1)
  [ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $R/gpurun_out/r04_cfg${N}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config $N --steps 20 --warmup 5 --no-cpu-baseline --no-pmc" > /dev/null
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); rm -rf /tmp/pmc_${N}_$i
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_${N}_$i -o pmc --output-format csv -- python $R/bench.py --config $N --steps 3 --warmup 1 --settle 2 --no-cpu-baseline --no-pmc > /tmp/pmc_${N}_$i.log 2>&1; echo "cfg $N pmc $set rc=$?"
  done
  python - $N $R <<'PY'
import csv, glob, json, sys, collections
n, root = int(sys.argv[1]), sys.argv[2]
bench = json.load(open(f"{root}/gpurun_out/r04_cfg{n}_bench.json"))
kname = bench["roofline"]["kernel"].split("<")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    for f in glob.glob(f"/tmp/pmc_{n}_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kname in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# the instantiation the timed steps launch = the one with the most dispatches (a counting pass may use another one)
for k, d in sorted(acc.items(), key=lambda kv: -len(kv[1].get("FETCH_SIZE", []))):
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fk = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); wk = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        rec = {"baseline_config": n, "kernel": k, "bytes_per_gpu": bench["config"]["bytes_per_gpu"],
               "command": f"rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python bench.py --config {n} --steps 3 --warmup 1 --settle 2 --no-cpu-baseline --no-pmc (scripts/gpu_r4_evidence.sh; one counter per pass)",
               "FETCH_SIZE_KB_mean": fk, "WRITE_SIZE_KB_mean": wk, "launches_averaged": len(d["FETCH_SIZE"]),
               "correction": "FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM); WRITE_SIZE as reported (uncalibrated)",
               "traffic_bytes_per_launch": int(2 * fk * 1024 + wk * 1024),
               "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"]}
        json.dump(rec, open(f"{root}/gpurun_out/r04_cfg{n}_pmc_traffic.json", "w"), indent=1)
        print("cfg", n, k, "traffic", rec["traffic_bytes_per_launch"], "algorithmic", rec["algorithmic_bytes_per_launch"])
        break
else:
    print("cfg", n, "no PMC rows for", kname)
b = bench
print("cfg", n, "value", b["value"], b["unit"], "ms/step", b["ms_per_step"], "kernel", b["roofline"]["kernel"], b["roofline"]["kernel_ms_avg"], "frac", b["roofline"]["frac"],
      "cpu1", b.get("cpu_baseline", {}).get("value"), "cpuN", b.get("cpu_baseline", {}).get("all_cores", {}).get("value"))
PY
done
# WRITE_SIZE calibration: a fill of exactly 1 GiB, the counter per launch against the bytes written
cd /tmp; rm -rf /tmp/pmc_cal
cat > /tmp/cal_fill.py <<'PY'
import torch
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for i in range(6):
    x.fill_(i)
torch.cuda.synchronize()
y = torch.empty((1 << 26, 2), dtype=torch.int64, device="cuda")      # 1 GiB of int64 pairs: copy_ reads 1 GiB and writes 1 GiB
z = torch.empty_like(y)
for i in range(4):
    z.copy_(y)
torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_cal -o cal --output-format csv -- python /tmp/cal_fill.py > /tmp/pmc_cal.log 2>&1; echo "calibration rc=$?"
python - $R <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_cal/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "WRITE_SIZE":
            acc[r["Kernel_Name"].split("(")[0][:90]].append(float(r["Counter_Value"]))
with open(f"{root}/gpurun_out/r04_write_size_calibration.txt", "w") as o:
    o.write("WRITE_SIZE calibration (rocprofv3 --kernel-trace --pmc WRITE_SIZE; scripts/gpu_r4_evidence.sh): kernels that write exactly 1 GiB = 1048576 KB per launch\n")
    for k, v in acc.items():
        o.write(f"{k}: launches {len(v)}, WRITE_SIZE mean {sum(v)/len(v):.0f} KB, ratio to 1048576 KB = {sum(v)/len(v)/1048576:.4f}\n")
print(open(f"{root}/gpurun_out/r04_write_size_calibration.txt").read())
PY
cd $R
timeout 900 python bench.py --total-gib 64 --steps 10 --warmup 2 --settle 3 --check-all-rows --no-pmc --no-cpu-baseline > gpurun_out/r04_cfg2_64gib_bench.json 2> gpurun_out/r04_cfg2_64gib_bench.err; echo "64 GiB rc=$?"; grep -o '"frac": [0-9.]*\|"kernel_ms_avg": [0-9.]*\|"checksums_equal": [a-z]*\|"corpus_checksum": "[0-9a-f]*"' gpurun_out/r04_cfg2_64gib_bench.json
