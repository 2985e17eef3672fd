# Round 5: rocprofv3 --kernel-trace --stats of the bench command WITHOUT the async leg (count + 40 settle + 5 warm-up + 20 timed launches: the
# average then covers what kernel_ms_avg times, plus the settle passes), for the five configs and the README IP pattern.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for N in 1 2 3 4 5; do
  rm -rf /tmp/prof_$N
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$N -o cfg$N -- python $R/bench.py --config $N --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async > /tmp/prof_$N.log 2>&1; echo "cfg $N stats rc=$?"
  db=$(find /tmp/prof_$N -name "*.db" | head -1)
  [ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $R/gpurun_out/r05_cfg${N}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config $N --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async" | sed -n 6p | cut -c1-150
  tail -1 /tmp/prof_$N.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench line of the profiled run: kernel_ms_avg', d['roofline']['kernel_ms_avg'], 'frac', d['roofline']['frac'])"
done
README_IP='(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
rm -rf /tmp/prof_fsm
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_fsm -o fsm -- python $R/bench.py --config 2 --pattern "$README_IP" --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async > /tmp/prof_fsm.log 2>&1
db=$(find /tmp/prof_fsm -name "*.db" | head -1)
[ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $R/gpurun_out/r05_fsm_readme_ip_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config 2 --pattern README_IP --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async" | sed -n 6p | cut -c1-150
for N in 1 2 3 4 5; do grep -A3 "launches in start order" $R/gpurun_out/r05_cfg${N}_kernel_stats.txt | cut -c1-160; done
