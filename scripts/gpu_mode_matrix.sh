# the transducer-kernel tests and one fuzz seed under the alternative runtime modes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
for v in "CXG_FSM_DEEP=1" "CXG_TICKETS=1" "CXG_NO_EPOCH=1" "CXG_NO_ZERO_COPY=1"; do
  echo "== $v"; env $v timeout 600 python -m pytest tests/test_gpu_fsm.py -x -q 2>&1 | tail -1
done
echo "== fuzz CXG_FSM_DEEP=1"; CXG_FSM_DEEP=1 timeout 900 python scripts/gpu_fuzz.py 81 1500 2>&1 | grep -v amdgpu | tail -3
echo "== fuzz look CXG_NO_EPOCH=1"; CXG_NO_EPOCH=1 FUZZ_LOOK=1 timeout 900 python scripts/gpu_fuzz.py 82 1500 2>&1 | grep -v amdgpu | tail -3
