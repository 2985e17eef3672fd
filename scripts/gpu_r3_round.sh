# Round 3, evidence call: the GPU test tier, smoke, then the per-config evidence and the 64 GiB single-GPU line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 1700 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r03_pytest_gpu.log 2>&1; echo pytest=$?; tail -14 gpurun_out/r03_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/gpu_r3_evidence.sh 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -40
bash scripts/gpu_r3_fsm_profile.sh 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -12
cd $R
timeout 900 python bench.py --total-gib 64 --gpus 1 --steps 10 --warmup 2 --settle 4 --check-all-rows --no-pmc > gpurun_out/r03_cfg2_64gib_bench.json 2> gpurun_out/r03_cfg2_64gib_bench.err; echo "64gib rc=$?"; tail -3 gpurun_out/r03_cfg2_64gib_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r03_cfg2_64gib_bench.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline'].get('all_rows_check'))"
