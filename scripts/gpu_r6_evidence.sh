# Round-6 evidence on the current tree (one box): per BASELINE config N = 1..5 (1 GiB, the bench.py command)
#   gpurun_out/r06_cfgN_bench.json          bench.py line (roofline incl. live PMC traffic + counters, cpu_baseline, async)
#   gpurun_out/r06_cfgN_kernel_stats.txt    rocprofv3 --kernel-trace --stats of the same command
# plus the general-DFA kernel on the README IP pattern, the default line (north_star: 64 GiB), the GPU tier and the device fuzz.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
timeout 500 python $R/bench.py > $R/gpurun_out/r06_bench_default.json 2> $R/gpurun_out/r06_bench_default.err; echo "default bench rc=$?"
for N in ${CFGS:-1 2 3 4 5}; do
  timeout 300 python $R/bench.py --config $N --steps 20 --warmup 5 --no-north-star > $R/gpurun_out/r06_cfg${N}_bench.json 2> $R/gpurun_out/r06_cfg${N}_bench.err; echo "cfg $N bench rc=$?"
  rm -rf /tmp/prof_$N
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$N -o cfg$N -- python $R/bench.py --config $N --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async > /tmp/prof_$N.log 2>&1; echo "cfg $N stats rc=$?"
  db=$(find /tmp/prof_$N -name "*.db" | head -1)
  [ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $R/gpurun_out/r06_cfg${N}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config $N --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async" > /dev/null
done
README_IP='(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
timeout 300 python $R/bench.py --config 2 --pattern "$README_IP" --steps 20 --warmup 5 --no-north-star > $R/gpurun_out/r06_fsm_readme_ip_bench.json 2> $R/gpurun_out/r06_fsm_readme_ip_bench.err; echo "fsm bench rc=$?"
rm -rf /tmp/prof_fsm
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_fsm -o fsm -- python $R/bench.py --config 2 --pattern "$README_IP" --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star > /tmp/prof_fsm.log 2>&1
db=$(find /tmp/prof_fsm -name "*.db" | head -1)
[ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $R/gpurun_out/r06_fsm_readme_ip_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config 2 --pattern README_IP --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async" > /dev/null
timeout 300 python $R/bench.py --config 2 --pattern '\b\d+\b' --steps 20 --warmup 5 --no-north-star > $R/gpurun_out/r06_fsm_word_digits_bench.json 2> $R/gpurun_out/r06_fsm_word_digits_bench.err; echo "word digits bench rc=$?"
rm -rf /tmp/prof_wd
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_wd -o fsm -- python $R/bench.py --config 2 --pattern '\b\d+\b' --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async > /tmp/prof_wd.log 2>&1
db=$(find /tmp/prof_wd -name "*.db" | head -1)
[ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $R/gpurun_out/r06_fsm_word_digits_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config 2 --pattern WORD_DIGITS --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async" > /dev/null
cd $R
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_cfg[1-5]_bench.json")) + ["gpurun_out/r06_fsm_readme_ip_bench.json", "gpurun_out/r06_fsm_word_digits_bench.json", "gpurun_out/r06_bench_default.json"]:
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r = b["roofline"]
    print(f.split("/")[-1], "value", b["value"], "ms/step", b["ms_per_step"], r["kernel"], r["kernel_ms_avg"], "frac", r["frac"], "traffic", r.get("traffic"), "alg", r["algorithmic_bytes_per_launch"],
          "cpu1", b.get("cpu_baseline", {}).get("value"), "async", (b.get("async") or {}).get("value"), "| north_star", (b.get("north_star") or {}).get("roofline", {}).get("frac"))
PY
timeout 250 python scripts/gpu_fuzz.py 101 260 > gpurun_out/r06_gpu_fuzz_general.txt 2>&1; tail -1 gpurun_out/r06_gpu_fuzz_general.txt | cut -c1-300
FUZZ_LOOK=1 timeout 200 python scripts/gpu_fuzz.py 102 180 > gpurun_out/r06_gpu_fuzz_look.txt 2>&1; tail -1 gpurun_out/r06_gpu_fuzz_look.txt | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
FUZZ_END=1 timeout 200 python scripts/gpu_fuzz.py 103 180 > gpurun_out/r06_gpu_fuzz_end.txt 2>&1; tail -1 gpurun_out/r06_gpu_fuzz_end.txt | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r06_pytest_gpu.log 2>&1; echo pytest=$?; tail -9 gpurun_out/r06_pytest_gpu.log | cut -c1-300
