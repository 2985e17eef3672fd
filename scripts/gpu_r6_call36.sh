# Round 6, call 36: the pair kernel (scan_teddy_pair.hip) — first run: Teddy parity tests, config 3 timing against the wave kernel.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrapped.py tests/test_zzz_gpu_fold.py -m gpu -q -k "teddy or edge_cases or reference_corpus or wrapped or fold or literal" > gpurun_out/r06_c36_pytest_teddy.log 2>&1; echo pytest=$?; tail -15 gpurun_out/r06_c36_pytest_teddy.log | cut -c1-400
timeout 200 python bench.py --config 3 --steps 20 --warmup 5 --no-north-star --no-cpu-baseline --no-pmc --no-async > gpurun_out/r06_c36_cfg3_pair.json 2> gpurun_out/r06_c36_cfg3_pair.err; echo "pair rc=$?"
CXG_NO_TEDDY_PAIR=1 timeout 200 python bench.py --config 3 --steps 20 --warmup 5 --no-north-star --no-cpu-baseline --no-pmc --no-async > gpurun_out/r06_c36_cfg3_wave.json 2> gpurun_out/r06_c36_cfg3_wave.err; echo "wave rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r06_c36_cfg3_pair.json", "gpurun_out/r06_c36_cfg3_wave.json"):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1]); r = b["roofline"]
        print(f.split("/")[-1], "value", b["value"], "ms/step", b["ms_per_step"], r["kernel"], r["kernel_ms_avg"], "frac", r["frac"])
    except Exception as e:
        print(f, "unreadable", e); print(open(f.replace(".json", ".err")).read()[-1500:])
PY
