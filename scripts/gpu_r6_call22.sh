# Round 6, call 22: mode 2 of the lean kernel with 1 024 rows per tile and four workgroups per CU against 2 048 rows / three (row-dense programs).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
PATS=('\b\d+\b' '\b\w+\b' '\d+' '[a-z]+\b')
for v in product rows1024 product rows1024; do
  if [ $v = product ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so; fi
  echo "== $v"; timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_c22_mode2_$v.txt | awk '{print $1, $(NF-7), $(NF-6), $(NF-5), $(NF-4), $(NF-3), $(NF-2)}'
done
