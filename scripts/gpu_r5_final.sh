# Round 5, last call: the GPU tier on the final tree, then config 4's bench line + rocprofv3 stats (the other configs' kernels are unchanged since r05_cfgN_*)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 560 python -m pytest tests -q -m gpu -x --durations=5 > gpurun_out/r05_final_pytest_gpu.log 2>&1; echo "pytest=$?"; tail -3 gpurun_out/r05_final_pytest_gpu.log
timeout 120 python bench.py --config 4 --steps 20 --warmup 5 --no-north-star 2> gpurun_out/r05_final_cfg4_bench.err | tail -1 > gpurun_out/r05_final_cfg4_bench.json; echo "bench4=$?"
python -c "
import json; d=json.loads(open('gpurun_out/r05_final_cfg4_bench.json').read()); r=d['roofline']; print('cfg4', d['value'], d['ms_per_step'], r['kernel_ms_avg'], r['frac'], r['traffic'])"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_4
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_4 -o cfg4 -- python $R/bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async > /tmp/prof_4.log 2>&1; echo "stats rc=$?"
db=$(find /tmp/prof_4 -name "*.db" | head -1)
[ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $R/gpurun_out/r05_final_cfg4_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async" | sed -n 6p | cut -c1-150
grep -A3 "launches in start order" $R/gpurun_out/r05_final_cfg4_kernel_stats.txt | cut -c1-160
