# Round 6, call 8: cxg_find / cxg_is_match, thread hygiene harness, async semantics — their tests; the default bench line with the fixed all-cores leg.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_find.py tests/test_boundary.py tests/test_gpu_async.py -m gpu -q -s > gpurun_out/r06_c8_pytest.log 2>&1; echo pytest=$?; grep -E "is_match over|passed|failed|Error" gpurun_out/r06_c8_pytest.log | tail -8 | cut -c1-300
timeout 600 python bench.py --no-north-star > gpurun_out/r06_c8_bench.json 2> gpurun_out/r06_c8_bench.err; echo bench=$?; python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_c8_bench.json"))
r = d["roofline"]; c = d.get("cpu_baseline", {})
print("1 GiB value", d["value"], "kernel", r["kernel_ms_avg"], "frac", r["frac"], "async", d.get("async", {}).get("value"))
print("cpu 1 thread", c.get("value"), "all cores", c.get("all_cores", {}).get("value"), "threads", c.get("all_cores", {}).get("cores"), "eff", c.get("all_cores", {}).get("scaling_efficiency"), c.get("all_cores", {}).get("runs_s"))
print(c.get("all_cores", {}).get("sample"))
PY
for N in 1 3 4; do timeout 300 python bench.py --config $N --steps 10 --warmup 3 --no-pmc --no-async 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['cpu_baseline']
print('cfg $N: kernel', d['roofline']['kernel_ms_avg'], 'frac', d['roofline']['frac'], '| cpu 1 thread', c['value'], 'all cores', c['all_cores']['value'], 'x', c['all_cores']['cores'], 'eff', c['all_cores']['scaling_efficiency'])"; done
