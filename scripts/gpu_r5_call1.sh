# Round 5, first device call: the whole GPU tier on the halo-carry tree (incl. the new golden-rows tests), the headline A/B
# (carry / no carry, unit sizes, occupancy, cache policy), FETCH_SIZE of product vs nocarry, the default bench line with north_star.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; V=$R/coregex_amd/variants
timeout 200 python -m pytest tests/test_gpu_fields.py tests/test_gpu_golden_rows.py -m gpu -q -x > gpurun_out/r05_c1_pytest_fields_golden.log 2>&1; echo pytest_fg=$?; tail -15 gpurun_out/r05_c1_pytest_fields_golden.log | cut -c1-400
{ timeout 60 python scripts/time_headline.py
  for v in nocarry t9 t11 t5 aux0; do CXG_LIB_PATH=$V/libcoregex_hip_$v.so timeout 60 python scripts/time_headline.py; done
  for occ in 4 5; do CXG_PF_OCC=$occ timeout 60 python scripts/time_headline.py; done
  CXG_PF_OCC=5 CXG_LIB_PATH=$V/libcoregex_hip_t9.so timeout 60 python scripts/time_headline.py
  timeout 60 python scripts/time_headline.py
  timeout 120 python scripts/time_headline.py 10 16
  CXG_LIB_PATH=$V/libcoregex_hip_nocarry.so timeout 120 python scripts/time_headline.py 10 16
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c1_headline_ab.txt; cat gpurun_out/r05_c1_headline_ab.txt
cd /tmp && export TMPDIR=/tmp
for v in product nocarry; do
  rm -rf /tmp/pmc_$v; L=""; [ $v != product ] && L=$V/libcoregex_hip_$v.so
  CXG_LIB_PATH=$L timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_$v -o pmc --output-format csv -- python $R/scripts/time_headline.py 3 > /dev/null 2>&1
  python - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]; vals = []
for f in glob.glob(f"/tmp/pmc_{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_scan_fields_pers" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE": vals.append(float(r["Counter_Value"]))
print(v, "FETCH_SIZE KB mean", sum(vals) / max(1, len(vals)), "launches", len(vals), "-> 2x bytes", 2 * 1024 * sum(vals) / max(1, len(vals)), "vs haystack", 1 << 30)
PY
done > $R/gpurun_out/r05_c1_fetch_size.txt 2>&1; cat $R/gpurun_out/r05_c1_fetch_size.txt
cd $R
timeout 600 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r05_c1_pytest_gpu.log 2>&1; echo pytest=$?; tail -12 gpurun_out/r05_c1_pytest_gpu.log | cut -c1-300
timeout 400 python bench.py > gpurun_out/r05_c1_bench_default.json 2> gpurun_out/r05_c1_bench_default.err; echo bench=$?; cut -c1-600 gpurun_out/r05_c1_bench_default.json; grep -o '"north_star".*' gpurun_out/r05_c1_bench_default.json | cut -c1-1500; tail -3 gpurun_out/r05_c1_bench_default.err
# this box's baseline for the other kernels (before this round's work on them)
{ timeout 200 python scripts/time_configs.py 1 3 4 5 2>&1 | grep -v amdgpu.ids
  timeout 200 python scripts/time_patterns.py '(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)' '\b\d+\.\d+\b' '\b\d+\b' '(?m)^\d+' 2>&1 | grep -v amdgpu.ids | sed 's/  */ /g'; } > gpurun_out/r05_c1_other_kernels_baseline.txt 2>&1; cat gpurun_out/r05_c1_other_kernels_baseline.txt | cut -c1-260
