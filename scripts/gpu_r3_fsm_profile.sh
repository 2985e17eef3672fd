# rocprofv3 kernel stats + PMC traffic of the transducer kernel on the README IP pattern and a word-boundary program
# (bench.py --pattern: ad-hoc workload on the config-2 corpus, no cpu_baseline) -> gpurun_out/r03_fsm_*.{json,txt}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
IP='(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
i=0
for PAT in "$IP" '\berror\b'; do
  i=$((i+1)); tag=$([ $i = 1 ] && echo readme_ip || echo word_boundary); cfg=$([ $i = 1 ] && echo 2 || echo 1)
  timeout 300 python $R/bench.py --pattern "$PAT" --synth-config $cfg --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $R/gpurun_out/r03_fsm_${tag}_bench.json 2>/dev/null; echo "$tag bench rc=$?"
  rm -rf /tmp/prof_f$i
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f$i -o f$i -- python $R/bench.py --pattern "$PAT" --synth-config $cfg --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > /tmp/prof_f$i.log 2>&1; echo "$tag stats rc=$?"
  db=$(find /tmp/prof_f$i -name "*.db" | head -1)
  [ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $R/gpurun_out/r03_fsm_${tag}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --pattern '$tag' --synth-config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-pmc" > /dev/null
  j=0
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    j=$((j+1)); rm -rf /tmp/pmc_f${i}_$j
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_f${i}_$j -o pmc --output-format csv -- python $R/bench.py --pattern "$PAT" --synth-config $cfg --config $cfg --steps 3 --warmup 1 --settle 2 --no-cpu-baseline --no-pmc > /dev/null 2>&1; echo "$tag pmc $set rc=$?"
  done
  python - $i $tag $R <<'PY'
import csv, glob, json, sys, collections
i, tag, root = sys.argv[1], sys.argv[2], sys.argv[3]
bench = json.load(open(f"{root}/gpurun_out/r03_fsm_{tag}_bench.json"))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for j in (1, 2):
    for f in glob.glob(f"/tmp/pmc_f{i}_{j}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_scan_fsm" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -len(kv[1].get("FETCH_SIZE", []))):
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fk = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); wk = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        rec = {"workload": bench["config"]["workload"], "kernel": k, "bytes_per_gpu": bench["config"]["bytes_per_gpu"],
               "FETCH_SIZE_KB_mean": fk, "WRITE_SIZE_KB_mean": wk, "traffic_bytes_per_launch": int(2 * fk * 1024 + wk * 1024),
               "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
               "correction": "FETCH_SIZE x2 (MI355X_MICROARCH.md HBM); WRITE_SIZE as reported"}
        json.dump(rec, open(f"{root}/gpurun_out/r03_fsm_{tag}_pmc_traffic.json", "w"), indent=1)
        print(tag, k, "traffic", rec["traffic_bytes_per_launch"], "algorithmic", rec["algorithmic_bytes_per_launch"])
        break
b = bench; print(tag, "value", b["value"], "kernel", b["roofline"]["kernel"], b["roofline"]["kernel_ms_avg"], "frac", b["roofline"]["frac"], "launches", b["roofline"]["launches_per_step"])
PY
done
