cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
{
echo "== persistent (default)"; timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "== grouped"; CXG_FIELDS_GROUPED=1 timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "== persistent"; timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
} > gpurun_out/r3u_modes.txt 2>&1; cat gpurun_out/r3u_modes.txt
