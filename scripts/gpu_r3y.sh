cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python -m pytest -m gpu -q -x tests/test_gpu_parity.py -k "submatch or 8gib or use_both or stops_the_wave" 2>&1 | tail -30 | cut -c1-260
