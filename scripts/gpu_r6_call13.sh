# Round 6, call 13: the lean kernel k_scan_fsml (class-indexed) against k_scan_fsm (CXG_FSM_NO_LEAN=1): times, instruction counts, the transducer's tests and fuzz.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
README_IP='(?:(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.){3}(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
PATS=("$README_IP" '\b\d+\b' '\b\d+\.\d+\b' '\d+\.\d+x?' '(?m)^\d+' 'a+b|b+a' '(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]' '[a-z]+://[^ ]+x?' 'GET|POST /[a-z]+')
echo "== lean"; timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c13_fsm_times_lean.txt | cut -c1-250
echo "== lean, class-indexed only"; CXG_FSM_NO_DIRECT=1 timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c13_fsm_times_lean_classidx.txt | cut -c1-250
echo "== k_scan_fsm"; CXG_FSM_NO_LEAN=1 timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c13_fsm_times_full.txt | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_fsm.py tests/test_gpu_text_anchor.py tests/test_zz_gpu_look_wider.py tests/test_gpu_golden_rows.py tests/test_gpu_nullable.py tests/test_zzzz_gpu_routing.py -m gpu -q -x > gpurun_out/r06_c13_pytest.log 2>&1; echo pytest=$?; tail -5 gpurun_out/r06_c13_pytest.log | cut -c1-300
timeout 200 python scripts/gpu_fuzz.py 641 200 > gpurun_out/r06_c13_gpu_fuzz_general.txt 2>&1; tail -1 gpurun_out/r06_c13_gpu_fuzz_general.txt | cut -c1-300
FUZZ_LOOK=1 timeout 200 python scripts/gpu_fuzz.py 642 150 > gpurun_out/r06_c13_gpu_fuzz_look.txt 2>&1; tail -1 gpurun_out/r06_c13_gpu_fuzz_look.txt | cut -c1-300
for p in 0 1; do
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$p && CXG_FSM_NO_DIRECT=1 timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_$p -o pmc --output-format csv -- python $R/scripts/time_patterns.py "${PATS[$p]}" > /tmp/pmc_$p.log 2>&1)
    f=$(find /tmp/pmc_$p -name "*counter_collection.csv" | head -1)
    python - "$f" "lean pattern $p" <<'PY' | tee -a gpurun_out/r06_c13_fsm_insts.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "k_scan" in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
tiles = (1 << 30) / 3840
for k, a in acc.items():
    print(sys.argv[2], k, "per wave-tile:", {c: round(sum(v) / len(v) / tiles, 1) for c, v in sorted(a.items()) if c.startswith("SQ_INSTS")}, "launches", len(a["SQ_WAVES"]))
PY
done
