# Round 6, call 23: mode 2 of the transducer kernels = 1 024 rows per tile (four workgroups per CU), the old mode (2 048 rows) behind it as mode 3: times, the GPU tier, fuzz with dense haystacks.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
PATS=('\b\d+\b' '\b\w+\b' '[a-z]+\b' '\b\w\b' '\b\d+\.\d+\b' '\d+\.\d+x?' 'a|b\b')
timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c23_dense_times.txt | awk '{print $1, $(NF-7), $(NF-6), $(NF-5), $(NF-4), $(NF-3), $(NF-2)}'
CXG_VERBOSE=1 timeout 100 python scripts/time_patterns.py '\b\w\b' 2>&1 | grep "cxg\]" | sort | uniq -c | head -8
timeout 1500 python -m pytest tests -m gpu -q --durations=3 > gpurun_out/r06_c23_pytest_gpu.log 2>&1; echo pytest=$?; tail -6 gpurun_out/r06_c23_pytest_gpu.log | cut -c1-300
timeout 250 python scripts/gpu_fuzz.py 691 300 > gpurun_out/r06_c23_gpu_fuzz_general.txt 2>&1; tail -1 gpurun_out/r06_c23_gpu_fuzz_general.txt | cut -c1-300
FUZZ_LOOK=1 timeout 250 python scripts/gpu_fuzz.py 692 300 > gpurun_out/r06_c23_gpu_fuzz_look.txt 2>&1; tail -1 gpurun_out/r06_c23_gpu_fuzz_look.txt | cut -c1-300
FUZZ_FEW=1 timeout 200 python scripts/gpu_fuzz.py 693 60 > gpurun_out/r06_c23_gpu_fuzz_few.txt 2>&1; tail -1 gpurun_out/r06_c23_gpu_fuzz_few.txt | cut -c1-300
for tag in word_digits; do
  P='\b\d+\b'
  timeout 300 python bench.py --config 2 --pattern "$P" --steps 20 --warmup 5 --no-north-star > gpurun_out/r06_fsm_${tag}_bench.json 2> gpurun_out/r06_fsm_${tag}_bench.err; echo "bench $tag rc=$?"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$tag && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o fsm -- python $R/bench.py --config 2 --pattern "$P" --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async > /tmp/prof_$tag.log 2>&1)
  db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocprof_summary.py $db gpurun_out/r06_fsm_${tag}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config 2 --pattern WORD_DIGITS --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async" | sed -n 6p | cut -c1-150
  python -c "
import json
d=json.loads(open('gpurun_out/r06_fsm_${tag}_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$tag', 'value', d['value'], 'kernel', r.get('kernel'), 'kernel_ms', r['kernel_ms_avg'], 'frac', r['frac'])"
done
