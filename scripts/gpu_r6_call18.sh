# Round 6, call 18: the north-star launch (64 GiB) on one box with the product, with the watchdog on s_memtime, and with round 5's scan_fields_wave.hip;
# a foreign kernel beside the persistent grid after the abort word; the tests of the persistent kernel.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
for v in product wdmemtime r05fields product; do
  if [ $v = product ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so; fi
  timeout 300 python bench.py --total-gib 64 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-north-star --no-async 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', '64 GiB kernel_ms', r['kernel_ms_avg'], 'frac', r['frac'], 'value', d['value'])" | tee -a gpurun_out/r06_c18_north_star_ab.txt
  timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', ' 1 GiB kernel_ms', r['kernel_ms_avg'], 'frac', r['frac'], 'value', d['value'])" | tee -a gpurun_out/r06_c18_north_star_ab.txt
done
unset CXG_LIB_PATH
for m in torchfill synthfill none; do timeout 200 python scripts/gpu_foreign_kernel.py $m 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_c18_foreign_kernel.txt | cut -c1-250; done
timeout 900 python -m pytest tests/test_gpu_watchdog.py tests/test_gpu_fields.py tests/test_gpu_literal.py tests/test_gpu_trio.py tests/test_gpu_shards_threads.py tests/test_gpu_async.py tests/test_gpu_u32_rows.py -m gpu -q -x > gpurun_out/r06_c18_pytest.log 2>&1; echo pytest=$?; tail -3 gpurun_out/r06_c18_pytest.log | cut -c1-300
