#!/usr/bin/env python3
"""Per-kernel means of the PMC passes collected by scripts/gpu_experiment.sh (gpurun_out/pmc_*/)."""
import csv, glob, collections, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "k_scan"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")

# --json OUT: write the traffic record bench.py reports as roofline.traffic (headline kernel only)
if "--json" in sys.argv:
    import json
    out = sys.argv[sys.argv.index("--json") + 1]
    for k, d in acc.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            fetch_kb = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); write_kb = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
            rec = {"kernel": k, "pattern": r"\d+\.\d+\.\d+\.\d+", "synth_config": 2, "bytes_per_gpu": 1 << 30, "digit_kernel": "6",
                   "command": "rocprofv3 --kernel-trace --pmc <set> -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline (scripts/gpu_pmc_traffic.sh; FETCH_SIZE and WRITE_SIZE in separate passes)",
                   "FETCH_SIZE_KB_mean": fetch_kb, "WRITE_SIZE_KB_mean": write_kb,
                   "correction": "FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM); WRITE_SIZE as reported (uncalibrated)",
                   "traffic_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
                   "counters": {c: sum(v) / len(v) for c, v in d.items()}}
            json.dump(rec, open(out, "w"), indent=1)
            print("wrote", out, rec["traffic_bytes_per_launch"])
            break
