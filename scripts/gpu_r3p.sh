cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_fields.py -q > gpurun_out/r3p_fields_tests.log 2>&1; echo "fields tests rc=$?"; tail -5 gpurun_out/r3p_fields_tests.log
{
echo "== grouped, two-level look-back (default)"; timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "== grouped, flat look-back (CXG_DEBUG=4)"; CXG_DEBUG=4 timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "== grouped, no look-back (CXG_DEBUG=2)"; CXG_DEBUG=2 timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "== no epochs (legacy memset path)"; CXG_NO_EPOCH=1 timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "== tickets"; CXG_TICKETS=1 timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
} > gpurun_out/r3p_modes.txt 2>&1; cat gpurun_out/r3p_modes.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3p_bench.json 2> gpurun_out/r3p_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r3p_bench.json')); print(d['value'], d['ms_per_step'], d['roofline'])"
