# quick loop: GPU parity tests + headline bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('IP', d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
