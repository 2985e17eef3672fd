# Round 5, call 13: config 4 — start-up stagger of the resident workgroups (phases out of step)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; V=$R/coregex_amd/variants
O=gpurun_out/r05_c13_cfg4.txt
{
  echo "product (EARLY=0, no stagger)"; timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
  for v in ccst16000 ccst32000 ccst48000; do echo "variant $v"; CXG_LIB_PATH=$V/libcoregex_hip_$v.so timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids; done
  echo "variant ccst32000, CXG_PROF=1"; CXG_PROF=1 CXG_LIB_PATH=$V/libcoregex_hip_ccst32000.so timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tail -2
  echo "product again"; timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids
} > $O 2>&1
cat $O
