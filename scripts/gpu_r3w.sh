cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_fields.py -q > gpurun_out/r3w_fields_tests.log 2>&1; echo "fields tests rc=$?"; tail -5 gpurun_out/r3w_fields_tests.log
{
timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
timeout 100 python scripts/time_modes.py '\d+:\d+:\d+' 2>&1 | grep -v amdgpu.ids | tail -1
timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
} > gpurun_out/r3w_modes.txt 2>&1; cat gpurun_out/r3w_modes.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3w_bench.json 2> gpurun_out/r3w_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r3w_bench.json')); print(d['value'], d['ms_per_step'], d['roofline'])"; tail -3 gpurun_out/r3w_bench.err
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_w; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_w -o cfg2 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > /tmp/prof_w.log 2>&1; echo "stats rc=$?"
db=$(find /tmp/prof_w -name "*.db" | head -1); [ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $R/gpurun_out/r3w_cfg2_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc" > /dev/null; head -20 $R/gpurun_out/r3w_cfg2_kernel_stats.txt
