# HBM traffic of the headline kernel: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (counters only + kernel trace)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/pmc_*
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --settle 2 --no-cpu-baseline --no-pmc > /dev/null 2>&1; echo "pmc $i rc=$?"
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py k_scan_chain_wave --json gpurun_out/pmc_traffic.json
