# Round 6, call 41: the pair kernel with the bank swizzle and plain ticket atomics — A/B with the wave kernel (rows must agree), PMC counts.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
O=gpurun_out/r06_c41_teddy_ab.txt; rm -f $O
timeout 200 python scripts/teddy_pair_ab.py cfg3 four fold short fat 2>&1 | grep -v amdgpu.ids | tee -a $O
CXG_NO_TEDDY_PAIR=1 timeout 200 python scripts/teddy_pair_ab.py cfg3 four fold short fat 2>&1 | grep -v amdgpu.ids | tee -a $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrapped.py tests/test_zzz_gpu_fold.py -m gpu -q -k "teddy or edge_cases or reference_corpus or wrapped or fold or literal" > gpurun_out/r06_c41_pytest_teddy.log 2>&1; echo pytest=$?; tail -3 gpurun_out/r06_c41_pytest_teddy.log | cut -c1-300
for v in pair wave; do
  if [ $v = wave ]; then export CXG_NO_TEDDY_PAIR=1; fi
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$v && timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$v -o pmc --output-format csv -- python $R/scripts/teddy_pair_ab.py cfg3 > /tmp/pmc_$v.log 2>&1)
    f=$(find /tmp/pmc_$v -name "*counter_collection.csv" | head -1)
    python - "$f" "$v" <<'PY' | tee -a gpurun_out/r06_c41_teddy_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "k_scan" in r["Kernel_Name"]: acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
tiles = (1 << 30) / 3840
for k, a in acc.items():
    print(sys.argv[2], k, "per wave-tile:", {c: round(sum(v) / len(v) / tiles, 1) for c, v in sorted(a.items())})
PY
  done
done
