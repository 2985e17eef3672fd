# Round 5, seventh device call: after the spin-sync revert — device fuzz, the tier (incl. captures of nullable patterns).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 250 python scripts/gpu_fuzz.py 101 260 > gpurun_out/r05_gpu_fuzz_general.txt 2>&1; tail -1 gpurun_out/r05_gpu_fuzz_general.txt | cut -c1-300; grep -c MISMATCH gpurun_out/r05_gpu_fuzz_general.txt
FUZZ_LOOK=1 timeout 200 python scripts/gpu_fuzz.py 102 180 > gpurun_out/r05_gpu_fuzz_look.txt 2>&1; tail -1 gpurun_out/r05_gpu_fuzz_look.txt | cut -c1-300
timeout 250 python scripts/gpu_fuzz.py 103 200 > gpurun_out/r05_gpu_fuzz_general2.txt 2>&1; tail -1 gpurun_out/r05_gpu_fuzz_general2.txt | cut -c1-300
timeout 700 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r05_pytest_gpu.log 2>&1; echo pytest=$?; tail -9 gpurun_out/r05_pytest_gpu.log | cut -c1-300
timeout 300 python bench.py --no-north-star --no-pmc --no-cpu-baseline > gpurun_out/r05_c7_bench.json 2> gpurun_out/r05_c7_bench.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_c7_bench.json"))
print("1 GiB", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], "async", d.get("async"))
PY
