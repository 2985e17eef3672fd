# Round 6, call 7: launch sections ordered on the device (no launch-to-completion mutex), C++ all-cores CPU leg, 64 GiB rocprofv3 stats — whole GPU tier,
# the default bench line, the 64 GiB kernel statistics, the foreign-kernel test, the FSM patterns.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 900 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r06_c7_pytest_gpu.log 2>&1; echo pytest=$?; tail -6 gpurun_out/r06_c7_pytest_gpu.log | cut -c1-300
timeout 600 python bench.py > gpurun_out/r06_c7_bench_default.json 2> gpurun_out/r06_c7_bench_default.err; echo bench=$?; python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_c7_bench_default.json"))
r = d["roofline"]; c = d.get("cpu_baseline", {}); ns = d.get("north_star", {})
print("1 GiB value", d["value"], "ms/step", d["ms_per_step"], "kernel", r["kernel_ms_avg"], "frac", r["frac"], "traffic", r.get("traffic"), "async", d.get("async", {}).get("value"))
print("cpu 1 thread", c.get("value"), "all cores", c.get("all_cores", {}).get("value"), "threads", c.get("all_cores", {}).get("cores"), "eff", c.get("all_cores", {}).get("scaling_efficiency"), c.get("all_cores", {}).get("runs_s"))
print("north_star", ns.get("value"), ns.get("roofline", {}).get("kernel_ms_avg"), ns.get("roofline", {}).get("frac"), ns.get("all_rows_check", {}).get("ok", ns.get("all_rows_check")))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_64
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_64 -o c64 -- python $R/bench.py --total-gib 64 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-north-star --no-async > /tmp/prof_64.log 2>&1; echo "64 GiB stats rc=$?"
db=$(find /tmp/prof_64 -name "*.db" | head -1)
[ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $R/gpurun_out/r06_cfg2_64gib_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --total-gib 64 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-north-star --no-async" | sed -n 1,12p | cut -c1-170
tail -1 /tmp/prof_64.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench line of the profiled 64 GiB run: kernel_ms_avg', d['roofline']['kernel_ms_avg'], 'frac', d['roofline']['frac'], 'value', d['value'])" | tee -a $R/gpurun_out/r06_cfg2_64gib_kernel_stats.txt
cd $R
timeout 300 python scripts/gpu_foreign_kernel.py > gpurun_out/r06_c7_foreign_kernel.txt 2>&1; tail -12 gpurun_out/r06_c7_foreign_kernel.txt | cut -c1-250
