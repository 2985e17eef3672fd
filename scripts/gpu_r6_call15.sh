# Round 6, call 15: the transducer kernels after the round's changes — times, the evidence files profiles/r06_fsm_{readme_ip,word_digits}_{bench.json,kernel_stats.txt},
# the whole GPU tier, device fuzz.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
README_IP='(?:(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.){3}(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
README_IP4='(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
PATS=("$README_IP" "$README_IP4" '\b\d+\b' '\b\d+\.\d+\b' '\d+\.\d+x?' '(?m)^\d+' 'a+b|b+a' '(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]' '\bGET\b|\bPOST\b' 'foo$|bar' '^\s+|\s+$')
timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c15_fsm_times.txt | cut -c1-250
for tag in readme_ip word_digits; do
  if [ $tag = readme_ip ]; then P="$README_IP4"; else P='\b\d+\b'; fi
  timeout 300 python bench.py --config 2 --pattern "$P" --steps 20 --warmup 5 --no-north-star > gpurun_out/r06_fsm_${tag}_bench.json 2> gpurun_out/r06_fsm_${tag}_bench.err; echo "bench $tag rc=$?"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$tag && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o fsm -- python $R/bench.py --config 2 --pattern "$P" --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async > /tmp/prof_$tag.log 2>&1)
  db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocprof_summary.py $db gpurun_out/r06_fsm_${tag}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config 2 --pattern $tag --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async" | sed -n 6p | cut -c1-150
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r06_fsm_${tag}_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$tag', 'value', d['value'], 'kernel', r.get('kernel'), 'kernel_ms', r['kernel_ms_avg'], 'frac', r['frac'], 'traffic', r.get('traffic'))"
done
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r06_c15_pytest_gpu.log 2>&1; echo pytest=$?; tail -12 gpurun_out/r06_c15_pytest_gpu.log | cut -c1-300
timeout 250 python scripts/gpu_fuzz.py 661 300 > gpurun_out/r06_c15_gpu_fuzz_general.txt 2>&1; tail -1 gpurun_out/r06_c15_gpu_fuzz_general.txt | cut -c1-300
FUZZ_LOOK=1 timeout 250 python scripts/gpu_fuzz.py 662 300 > gpurun_out/r06_c15_gpu_fuzz_look.txt 2>&1; tail -1 gpurun_out/r06_c15_gpu_fuzz_look.txt | cut -c1-300
FUZZ_END=1 timeout 200 python scripts/gpu_fuzz.py 663 200 > gpurun_out/r06_c15_gpu_fuzz_end.txt 2>&1; tail -1 gpurun_out/r06_c15_gpu_fuzz_end.txt | cut -c1-300
FUZZ_FEW=1 timeout 200 python scripts/gpu_fuzz.py 664 60 > gpurun_out/r06_c15_gpu_fuzz_few.txt 2>&1; tail -1 gpurun_out/r06_c15_gpu_fuzz_few.txt | cut -c1-300
