# Round 6, call 4: ticketed persistent kernel (units claimed from per-XCD counters, block-level decoupled look-back) — tests, bench, A/B against the static variant.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fields.py tests/test_gpu_literal.py tests/test_gpu_watchdog.py tests/test_gpu_shards_threads.py tests/test_gpu_async.py tests/test_gpu_u32_rows.py tests/test_gpu_trio.py -m gpu -x -q > gpurun_out/r06_c4_pytest_pers.log 2>&1; echo pytest=$?; tail -6 gpurun_out/r06_c4_pytest_pers.log | cut -c1-300
for v in base static; do
  if [ $v = base ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so; fi
  echo "== $v"; timeout 300 python bench.py --steps 20 --warmup 5 --no-north-star --no-pmc --no-cpu-baseline > gpurun_out/r06_c4_bench_$v.json 2> gpurun_out/r06_c4_bench_$v.err; python - <<PY
import json
d = json.load(open("gpurun_out/r06_c4_bench_$v.json"))
print("1 GiB", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], "async", d.get("async"))
PY
  timeout 300 python bench.py --total-gib 16 --steps 10 --warmup 2 --no-north-star --no-pmc --no-cpu-baseline > gpurun_out/r06_c4_bench16_$v.json 2> gpurun_out/r06_c4_bench16_$v.err; python - <<PY
import json
d = json.load(open("gpurun_out/r06_c4_bench16_$v.json"))
print("16 GiB", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"])
PY
done
unset CXG_LIB_PATH
timeout 200 python scripts/time_configs.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c4_configs.txt | cut -c1-200
