# Round 6, call 59: where the pair kernel's extra HBM fetches come from — FETCH_SIZE of builds without the verifier's requests (pabl64) and with
# other cache policies on the window loads (pw: 1 sc0, 2 nt, 16 sc1, 17 both) or on the requests (pv).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
O=gpurun_out/r06_c59_pair_fetch.txt; rm -f $O
for v in product pabl64 pw1 pw2 pw16 pw17 pv1 pv2 pv16 pv17; do
  if [ $v = product ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so; fi
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_f && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o pmc --output-format csv -- python $R/scripts/pair_abl_time.py > /tmp/pmc_f.log 2>&1)
  f=$(find /tmp/pmc_f -name "*counter_collection.csv" | head -1)
  python - "$f" "$v" <<'PY' | tee -a $O
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_scan_teddy" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE": acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print(sys.argv[2], k, "FETCH_SIZE KB mean", round(sum(v) / len(v), 1), "x2 / GiB", round(2 * sum(v) / len(v) * 1024 / (1 << 30), 4), "launches", len(v))
PY
  grep kernel_ms /tmp/pmc_f.log | cut -c1-120 | tee -a $O
done
