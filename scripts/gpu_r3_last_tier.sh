cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu.log 2>&1; echo pytest=$?; grep -E "^FAILED|passed|failed" gpurun_out/r03_pytest_gpu.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --config 5 --steps 20 --warmup 5 --no-pmc 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
