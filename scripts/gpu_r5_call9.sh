# Round 5, call 9: where a workgroup of the char-class kernel spends its time (CXG_PROF: s_memtime ticks of wave 0 per phase)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; V=$R/coregex_amd/variants
O=gpurun_out/r05_c9_cfg4_phases.txt
{
  echo "product, CXG_PROF=1 (first 4 lines: count-only launches; then launches with rows)"
  CXG_PROF=1 timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "variant ccd4h (4 windows in flight, two-halves extraction)"
  CXG_PROF=1 CXG_LIB_PATH=$V/libcoregex_hip_ccd4h.so timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  echo "product, tickets instead of static groups"
  CXG_TICKETS=1 CXG_PROF=1 timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids | tail -3
} > $O 2>&1
cat $O
