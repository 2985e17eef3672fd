cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
{ echo "product"; timeout 120 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
  echo "variant nt (nontemporal row stores)"; CXG_LIB_PATH=$GRAFT_REPO_ROOT/coregex_amd/variants/libcoregex_hip_nt.so timeout 120 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
  echo grouped; CXG_NO_PERSIST=1 timeout 120 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
  echo "64 GiB, persistent"; timeout 300 python bench.py --total-gib 64 --steps 5 --warmup 1 --settle 2 --no-cpu-baseline --no-pmc 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms_avg": [0-9.]*\|"frac": [0-9.]*'
  echo "64 GiB, grouped"; CXG_NO_PERSIST=1 timeout 300 python bench.py --total-gib 64 --steps 5 --warmup 1 --settle 2 --no-cpu-baseline --no-pmc 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms_avg": [0-9.]*\|"frac": [0-9.]*'
  echo "64 GiB, nt"; CXG_LIB_PATH=$GRAFT_REPO_ROOT/coregex_amd/variants/libcoregex_hip_nt.so timeout 300 python bench.py --total-gib 64 --steps 5 --warmup 1 --settle 2 --no-cpu-baseline --no-pmc 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms_avg": [0-9.]*\|"frac": [0-9.]*'
} > gpurun_out/r04_pers_64g.txt 2>&1; cat gpurun_out/r04_pers_64g.txt
