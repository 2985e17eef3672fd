# Round 5, fifth device call: own occupancy count for the persistent instantiations, 50 ms wait limit; char-class A/B; the tier.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; V=$R/coregex_amd/variants
{ echo "product"; CXG_VERBOSE=1 timeout 200 python scripts/time_configs.py 1 4 2>&1 | grep -v "XCD\|waves;\|units waited"
  for v in ccnosw ccr4 ccsw2; do echo "variant $v"; CXG_LIB_PATH=$V/libcoregex_hip_$v.so timeout 100 python scripts/time_configs.py 4; done
  echo "product again"; timeout 100 python scripts/time_configs.py 4
  echo "literals on 1 GiB of config 2"; timeout 100 python scripts/time_patterns.py 'GET' 'HTTP/' 'error' 2>&1 | sed 's/  */ /g'
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c5_configs.txt; cat gpurun_out/r05_c5_configs.txt | cut -c1-330
{ for m in torchfill synthfill; do CXG_VERBOSE=1 timeout 120 python scripts/gpu_foreign_kernel.py $m 2>&1 | grep -v "XCD\|waves;\|units waited\|amdgpu.ids\|workgroups per CU" | tail -3; done
} > gpurun_out/r05_c5_foreign_kernel.txt 2>&1; cat gpurun_out/r05_c5_foreign_kernel.txt | cut -c1-300
timeout 700 python -m pytest tests -m gpu -q --durations=5 -x > gpurun_out/r05_c5_pytest_gpu.log 2>&1; echo pytest=$?; tail -15 gpurun_out/r05_c5_pytest_gpu.log | cut -c1-400
