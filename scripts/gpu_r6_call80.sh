# Round 6, call 80: the pair kernel's groups tapering off (8 tiles per wave, then one group of 6, 4 and 2 per CU) against the first form (CXG_PAIR_TAPER=0).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; export CXG_PAIR_MIN_BYTES=0
O=gpurun_out/r06_c80_pair_taper.txt; rm -f $O
for t in 1 0 1 0; do CXG_PAIR_TAPER=$t timeout 120 python scripts/pair_sizes_big.py 2>&1 | grep -v amdgpu.ids | tee -a $O; done
bash scripts/gpu_r6_call51.sh
timeout 600 python -m pytest tests/test_gpu_teddy_pair.py -m gpu -q 2>&1 | tail -2 | cut -c1-300
