# Round 6, call 79: config 3 at 16 GiB on one GPU (the pair kernel's fixed costs — 15 us of tables, the last groups — spread over 16 times the bytes), rows checked.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --config 3 --total-gib 16 --steps 10 --warmup 3 --no-north-star --no-cpu-baseline --no-pmc --no-async --check-all-rows > $R/gpurun_out/r06_cfg3_16gib_bench.json 2> $R/gpurun_out/r06_cfg3_16gib_bench.err; echo rc=$?
CXG_NO_TEDDY_PAIR=1 timeout 600 python $R/bench.py --config 3 --total-gib 16 --steps 10 --warmup 3 --no-north-star --no-cpu-baseline --no-pmc --no-async > $R/gpurun_out/r06_cfg3_16gib_wave_kernel_bench.json 2> /dev/null; echo rc=$?
cd $R; python - <<'PY'
import json
for f in ("gpurun_out/r06_cfg3_16gib_bench.json", "gpurun_out/r06_cfg3_16gib_wave_kernel_bench.json"):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1]); r = b["roofline"]
        print(f.split("/")[-1], "value", b["value"], "ms/step", b["ms_per_step"], r["kernel"], r["kernel_ms_avg"], "frac", r["frac"], b.get("rows_checked"), str(b.get("config"))[:120])
    except Exception as e:
        print(f, "unreadable", e); print(open(f.replace(".json", ".err")).read()[-800:] if f.endswith("bench.json") else "")
PY
