# after k_scan_trio_wave: GPU tier on the final tree, config 5 evidence, instruction counters
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 1700 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r03_pytest_gpu.log 2>&1; echo pytest=$?; grep -E "^FAILED|passed|failed" gpurun_out/r03_pytest_gpu.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
CFGS="5" bash scripts/gpu_r3_evidence.sh 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -6
cd $R
bash scripts/gpu_pmc_configs.sh > gpurun_out/r03_all_configs_pmc_counters.txt 2>&1; grep -A8 "trio_wave" gpurun_out/r03_all_configs_pmc_counters.txt | head -12
timeout 300 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo bench=$?; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['all_cores']['value'])"
