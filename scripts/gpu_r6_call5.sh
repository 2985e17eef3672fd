# Round 6, call 5: ticketed persistent kernel with 64 workgroup-keyed counters — tests, then A/B of the tail geometry against the static variant.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fields.py tests/test_gpu_literal.py tests/test_gpu_watchdog.py tests/test_gpu_shards_threads.py tests/test_gpu_async.py tests/test_gpu_u32_rows.py tests/test_gpu_trio.py -m gpu -x -q > gpurun_out/r06_c6_pytest_pers.log 2>&1; echo pytest=$?; tail -3 gpurun_out/r06_c6_pytest_pers.log | cut -c1-300
for v in base t10 t00 static; do
  if [ $v = base ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-north-star --no-pmc --no-cpu-baseline > gpurun_out/r06_c6_bench_$v.json 2> gpurun_out/r06_c6_bench_$v.err
  timeout 300 python bench.py --total-gib 16 --steps 10 --warmup 2 --no-north-star --no-pmc --no-cpu-baseline > gpurun_out/r06_c6_bench16_$v.json 2> gpurun_out/r06_c6_bench16_$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r06_c6_bench_$v.json")); e = json.load(open("gpurun_out/r06_c6_bench16_$v.json"))
print("$v", "1 GiB kernel", d["roofline"]["kernel_ms_avg"], "frac", d["roofline"]["frac"], "value", d["value"], "async", d["async"]["value"], "| 16 GiB kernel", e["roofline"]["kernel_ms_avg"], "frac", e["roofline"]["frac"])
PY
done 2>&1 | tee gpurun_out/r06_c6_tail_ab.txt
