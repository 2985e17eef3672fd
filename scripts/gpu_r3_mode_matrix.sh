# round 3: the transducer / fields GPU tests and fuzz seeds under the alternative runtime modes (evidence: profiles/r03_gpu_mode_matrix.txt)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
{
for v in "CXG_FSM_DEEP=1" "CXG_TICKETS=1" "CXG_NO_EPOCH=1" "CXG_NO_ZERO_COPY=1"; do
  echo "== $v: tests/test_gpu_fsm.py tests/test_gpu_fields.py"; env $v timeout 600 python -m pytest tests/test_gpu_fsm.py tests/test_gpu_fields.py -x -q 2>&1 | tail -1
done
echo "== fuzz few symbols, CXG_FSM_DEEP=1"; CXG_FSM_DEEP=1 FUZZ_FEW=1 timeout 900 python scripts/gpu_fuzz.py 83 200 2>&1 | grep -v amdgpu | tail -3
echo "== fuzz few symbols, CXG_TICKETS=1"; CXG_TICKETS=1 FUZZ_FEW=1 timeout 900 python scripts/gpu_fuzz.py 84 200 2>&1 | grep -v amdgpu | tail -3
echo "== fuzz few symbols + look-around"; FUZZ_LOOK=1 FUZZ_FEW=1 timeout 900 python scripts/gpu_fuzz.py 85 250 2>&1 | grep -v amdgpu | tail -3
echo "== fuzz general, CXG_NO_EPOCH=1"; CXG_NO_EPOCH=1 timeout 900 python scripts/gpu_fuzz.py 86 600 2>&1 | grep -v amdgpu | tail -3
echo "== fuzz look-around"; FUZZ_LOOK=1 timeout 900 python scripts/gpu_fuzz.py 87 800 2>&1 | grep -v amdgpu | tail -3
} > gpurun_out/r03_gpu_mode_matrix.txt 2>&1; cat gpurun_out/r03_gpu_mode_matrix.txt
