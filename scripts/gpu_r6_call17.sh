# Round 6, call 17: lean kernel with the flat epilogue and carried tile offsets; the persistent kernel's watchdog on s_memrealtime (5 ms); the default bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
README_IP4='(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
PATS=("$README_IP4" '\b\d+\b' '\b\d+\.\d+\b' '\d+\.\d+x?' '(?m)^\d+' 'a+b|b+a' '(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]' '\bGET\b|\bPOST\b')
timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c17_fsm_times.txt | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_fsm.py tests/test_gpu_text_anchor.py tests/test_zz_gpu_look_wider.py tests/test_gpu_golden_rows.py tests/test_gpu_watchdog.py tests/test_gpu_shards_threads.py tests/test_gpu_async.py tests/test_boundary.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r06_c17_pytest.log 2>&1; echo pytest=$?; tail -3 gpurun_out/r06_c17_pytest.log | cut -c1-300
timeout 200 python scripts/gpu_fuzz.py 671 200 > gpurun_out/r06_c17_gpu_fuzz_general.txt 2>&1; tail -1 gpurun_out/r06_c17_gpu_fuzz_general.txt | cut -c1-300
FUZZ_LOOK=1 timeout 200 python scripts/gpu_fuzz.py 672 150 > gpurun_out/r06_c17_gpu_fuzz_look.txt 2>&1; tail -1 gpurun_out/r06_c17_gpu_fuzz_look.txt | cut -c1-300
timeout 200 python scripts/gpu_foreign_kernel.py > gpurun_out/r06_c17_foreign_kernel.txt 2>&1; tail -8 gpurun_out/r06_c17_foreign_kernel.txt | cut -c1-250
timeout 600 python bench.py > gpurun_out/r06_c17_bench_default.json 2> gpurun_out/r06_c17_bench_default.err; echo bench=$?
python -c "
import json
d=json.loads(open('gpurun_out/r06_c17_bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms_avg'], 'frac', r['frac'], 'traffic/B_alg', round(r['traffic']/r['algorithmic_bytes_per_launch'],4) if r.get('traffic') else None)
print('async', d.get('async',{}).get('value'), 'north_star', {k:d['north_star'].get(k) for k in ('frac','kernel_ms_avg','rows_checked')} if 'north_star' in d else None)
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('all_cores',{}).get('value'))"
