# Round 6, call 9: phases of the transducer kernel on dense-row look-around programs; find/is_match + hygiene tests again; bench with the effective-core detection.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
timeout 600 python -m pytest tests/test_gpu_find.py tests/test_boundary.py -m gpu -q -s > gpurun_out/r06_c9_pytest.log 2>&1; echo pytest=$?; grep -E "is_match over|passed|failed|Error" gpurun_out/r06_c9_pytest.log | tail -6 | cut -c1-300
CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_fsmprof.so CXG_PROF=1 timeout 200 python scripts/time_patterns.py '\b\d+\b' '\b\d+\.\d+\b' 2>&1 | grep -E "CXG_PROF|kernel_ms" | cut -c1-300 | tail -6 > gpurun_out/r06_c9_fsm_phases_dense.txt; cat gpurun_out/r06_c9_fsm_phases_dense.txt
timeout 300 python bench.py --no-north-star --no-pmc --no-async --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['cpu_baseline']
print('cfg 2: cpu 1 thread', c['value'], 'all cores', c['all_cores']['value'], 'x', c['all_cores']['cores'], 'eff', c['all_cores']['scaling_efficiency'], 'affinity', c['host_threads_in_affinity_mask'], 'quota', c['cgroup_cpu_quota_cores'], 'effective', c['effective_cores_by_spin_test'])"
