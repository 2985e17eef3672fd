# Round 6, call 51: the pair kernel back on group-level look-back (one barrier per group) with a wave's tiles consecutive; A/B, stress, tests, instruction counts.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
T=${TAG:-c82}
O=gpurun_out/r06_${T}_pair.txt; rm -f $O
timeout 200 python scripts/teddy_pair_ab.py cfg3 four fold 2>&1 | grep -v amdgpu.ids | tee -a $O
timeout 300 python scripts/gpu_stress_pair.py 20000 2>&1 | grep -v amdgpu.ids | tee -a $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrapped.py tests/test_zzz_gpu_fold.py -m gpu -q -k "teddy or edge_cases or reference_corpus or wrapped or fold or literal" > gpurun_out/r06_${T}_pytest_teddy.log 2>&1; echo pytest=$?; grep -E "AssertionError: |passed|failed" gpurun_out/r06_${T}_pytest_teddy.log | cut -c1-600
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_p && timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_p -o pmc --output-format csv -- python $R/scripts/teddy_pair_ab.py cfg3 > /tmp/pmc_p.log 2>&1)
  f=$(find /tmp/pmc_p -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY' | tee -a $O
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "k_scan" in r["Kernel_Name"]: acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
tiles = (1 << 30) / 3840
for k, a in acc.items():
    print(k, "per wave-tile:", {c: round(sum(v) / len(v) / tiles, 1) for c, v in sorted(a.items())})
PY
done
