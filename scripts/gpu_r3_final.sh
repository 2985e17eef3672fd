# Round 3, final call: GPU tier + smoke on the final tree, evidence of the configs whose kernels changed after gpu_r3_round.sh
# (config 3: literal kernel; transducer profiles), the default bench line, instruction counters.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 1700 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r03_pytest_gpu.log 2>&1; echo pytest=$?; tail -10 gpurun_out/r03_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
CFGS="3 1" bash scripts/gpu_r3_evidence.sh 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -12
bash scripts/gpu_r3_fsm_profile.sh 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -6
cd $R
timeout 300 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo bench=$?; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline']['value'], d['cpu_baseline']['all_cores'])"
bash scripts/gpu_pmc_configs.sh > gpurun_out/r03_all_configs_pmc_counters.txt 2>&1; grep -A3 "teddy_wave" gpurun_out/r03_all_configs_pmc_counters.txt | head -8
