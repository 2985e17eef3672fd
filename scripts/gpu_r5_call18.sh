# Round 5, call 18: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1) — every workgroup's first scalar loads
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
O=gpurun_out/r05_c18_dev_kernarg.txt
{
  echo "default"; CXG_PROF=1 timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tail -2
  echo "HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 CXG_PROF=1 timeout 100 python scripts/time_configs.py 4 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tail -2
  echo "default, all configs"; timeout 200 python scripts/time_configs.py 2>&1 | grep -v amdgpu.ids
  echo "HIP_FORCE_DEV_KERNARG=1, all configs"; HIP_FORCE_DEV_KERNARG=1 timeout 200 python scripts/time_configs.py 2>&1 | grep -v amdgpu.ids
} > $O 2>&1
cat $O
