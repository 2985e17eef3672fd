# Round 6, call 77: a foreign kernel beside the PAIR kernel's persistent grid (claimed groups: forward progress by construction) and, for comparison,
# beside the headline's persistent grid (co-residency + watchdog).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
O=gpurun_out/r06_c77_foreign_kernel_pair.txt; rm -f $O
for m in none torchfill synthfill; do timeout 300 python scripts/gpu_foreign_kernel.py $m literals 2>&1 | grep -v amdgpu.ids | tee -a $O; done
for m in none torchfill synthfill; do timeout 300 python scripts/gpu_foreign_kernel.py $m 2>&1 | grep -v amdgpu.ids | tee -a $O; done
