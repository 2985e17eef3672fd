# Last device call of round 4: device fuzz (general, look-around), smoke(), the whole GPU tier and the default bench line on the tree that is left behind.
# (CFGS=2 scripts/gpu_r4_evidence.sh + the 64 GiB line ran in the previous call of this script's first version: profiles/r04_final_cfg2_*.)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 100 python scripts/gpu_fuzz.py 93 300 > gpurun_out/r04_gpu_fuzz_general_final.txt 2>&1; tail -1 gpurun_out/r04_gpu_fuzz_general_final.txt | cut -c1-300
FUZZ_LOOK=1 timeout 100 python scripts/gpu_fuzz.py 94 200 > gpurun_out/r04_gpu_fuzz_look_final.txt 2>&1; tail -1 gpurun_out/r04_gpu_fuzz_look_final.txt | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r04_pytest_gpu.log 2>&1; echo pytest=$?; tail -9 gpurun_out/r04_pytest_gpu.log | cut -c1-300
timeout 150 python bench.py --steps 20 --warmup 3 > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo bench=$?; cut -c1-420 gpurun_out/r04_bench_final.json
