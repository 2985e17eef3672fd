cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_fields.py -q > gpurun_out/r3r_fields_tests.log 2>&1; echo "fields tests rc=$?"; tail -5 gpurun_out/r3r_fields_tests.log
{
echo "== persistent (default)"; timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "== grouped (CXG_FIELDS_GROUPED=1)"; CXG_FIELDS_GROUPED=1 timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "== persistent again"; timeout 100 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
for p in '\d+:\d+:\d+' '\d+\.\d+'; do timeout 100 python scripts/time_modes.py "$p" 2>&1 | grep -v amdgpu.ids | tail -1; done
} > gpurun_out/r3r_modes.txt 2>&1; cat gpurun_out/r3r_modes.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3r_bench.json 2> gpurun_out/r3r_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r3r_bench.json')); print(d['value'], d['ms_per_step'], d['roofline'])"
