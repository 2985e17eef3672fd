# Round 6, call 11: branch-free first 16 steps of the reverse walks (fsm_match_start16) against the loop alone: times, instruction counts, tests of the transducer kernel.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
README_IP='(?:(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.){3}(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
PATS=("$README_IP" '\b\d+\b' '\b\d+\.\d+\b' '\d+\.\d+x?' '(?m)^\d+' 'a+b|b+a' '(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]' 'https?://[^ ]+x?' '[a-z]+@[a-z]+\.com')
echo "== fast starts"; CXG_FSM_NO_DIRECT=1 timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c11_fsm_times_fast.txt | cut -c1-250
echo "== loop only"; CXG_FSM_NO_DIRECT=1 CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_slowstarts.so timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c11_fsm_times_loop.txt | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_fsm.py tests/test_gpu_text_anchor.py tests/test_zz_gpu_look_wider.py tests/test_gpu_golden_rows.py -m gpu -q -x > gpurun_out/r06_c11_pytest.log 2>&1; echo pytest=$?; tail -3 gpurun_out/r06_c11_pytest.log | cut -c1-300
for v in fast loop; do
  if [ $v = loop ]; then export CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_slowstarts.so; else unset CXG_LIB_PATH; fi
  for p in 0 1; do
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$v$p && CXG_FSM_NO_DIRECT=1 timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_$v$p -o pmc --output-format csv -- python $R/scripts/time_patterns.py "${PATS[$p]}" > /tmp/pmc_$v$p.log 2>&1)
    f=$(find /tmp/pmc_$v$p -name "*counter_collection.csv" | head -1)
    python - "$f" "$v pattern $p" <<'PY' | tee -a gpurun_out/r06_c11_fsm_insts.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_scan_fsm" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
tiles = (1 << 30) / 3840
print(sys.argv[2], "per wave-tile:", {k: round(sum(v) / len(v) / tiles, 1) for k, v in sorted(acc.items()) if k.startswith("SQ_INSTS")}, "launches", len(acc["SQ_WAVES"]))
PY
  done
done
