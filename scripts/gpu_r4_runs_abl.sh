cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
IP='(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
{ for d in 0 16 32; do echo "CXG_DEBUG=$d"; CXG_DEBUG=$d timeout 100 python scripts/time_patterns.py "$IP" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c170-; done
} > gpurun_out/r04_runs_abl.txt 2>&1; cat gpurun_out/r04_runs_abl.txt
