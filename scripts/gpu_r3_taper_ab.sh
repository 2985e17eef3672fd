cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_fields.py -q -x > gpurun_out/r3t_fields_tests.log 2>&1; echo "fields tests rc=$?"; tail -5 gpurun_out/r3t_fields_tests.log
{
for cfg in "0 2048" "2 1024" "2 2048" "2 3072" "4 2048" "4 3072" "1 2048" "0 2048" "2 2048"; do set -- $cfg
  echo "CXG_TAPER_TPW=$1 CXG_TAPER_GROUPS=$2: $(CXG_TAPER_TPW=$1 CXG_TAPER_GROUPS=$2 timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1)"
done
} > gpurun_out/r3t_taper.txt 2>&1; cat gpurun_out/r3t_taper.txt
