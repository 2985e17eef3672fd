cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
{ echo "pipelined (default)"; timeout 300 python scripts/time_host_path_c.py 2>&1 | grep MiB; echo "CXG_NO_PIPELINE=1"; CXG_NO_PIPELINE=1 timeout 300 python scripts/time_host_path_c.py 2>&1 | grep MiB; } > gpurun_out/r3s_host_path_c.txt; cat gpurun_out/r3s_host_path_c.txt
