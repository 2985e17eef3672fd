cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
{
for m in "CXG_FIELDS_GROUPED=1" "CXG_FIELDS_GROUPED=1 CXG_DEBUG=4" "CXG_FIELDS_GROUPED=1 CXG_DEBUG=2"; do echo "== $m"; env $m CXG_PROF=1 timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -4; done
} > gpurun_out/r3t_prof.txt 2>&1; cat gpurun_out/r3t_prof.txt
