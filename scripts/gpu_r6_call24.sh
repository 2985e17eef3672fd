# Round 6, call 24: final tree — end-of-text programs on their own image sizes, occupancy targets by image size; the GPU tier; fuzz; smoke; the default bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
README_IP4='(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
PATS=("$README_IP4" '\b\d+\b' '\b\d+\.\d+\b' '\d+\.\d+x?' '(?m)^\d+' 'a+b|b+a' '(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]' '\bGET\b|\bPOST\b' 'foo$|bar' 'x\z|error' '\d+$|warn')
timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c24_fsm_times.txt | awk '{print $1, $(NF-7), $(NF-6), $(NF-5), $(NF-4), $(NF-3), $(NF-2)}' | cut -c1-120
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q --durations=3 > gpurun_out/r06_c24_pytest_gpu.log 2>&1; echo pytest=$?; tail -4 gpurun_out/r06_c24_pytest_gpu.log | cut -c1-300
FUZZ_END=1 timeout 200 python scripts/gpu_fuzz.py 701 250 > gpurun_out/r06_c24_gpu_fuzz_end.txt 2>&1; tail -1 gpurun_out/r06_c24_gpu_fuzz_end.txt | cut -c1-300
FUZZ_LOOK=1 timeout 200 python scripts/gpu_fuzz.py 702 250 > gpurun_out/r06_c24_gpu_fuzz_look.txt 2>&1; tail -1 gpurun_out/r06_c24_gpu_fuzz_look.txt | cut -c1-300
timeout 200 python scripts/gpu_fuzz.py 703 250 > gpurun_out/r06_c24_gpu_fuzz_general.txt 2>&1; tail -1 gpurun_out/r06_c24_gpu_fuzz_general.txt | cut -c1-300
timeout 600 python bench.py > gpurun_out/r06_c24_bench_default.json 2> gpurun_out/r06_c24_bench_default.err; echo bench=$?
python -c "
import json
d=json.loads(open('gpurun_out/r06_c24_bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']; ns=d['north_star']
print('value', d['value'], 'kernel_ms', r['kernel_ms_avg'], 'frac', r['frac'], 'async', d['async']['value'], 'north_star', ns['roofline']['frac'], ns['roofline']['kernel_ms_avg'], ns['all_rows_check'].get('checksums_equal'))"
