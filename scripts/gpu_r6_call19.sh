# Round 6, call 19: the lean kernel aiming at five workgroups per CU (96 VGPRs) against four.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
README_IP4='(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
PATS=("$README_IP4" '\b\d+\b' '\b\d+\.\d+\b' '\d+\.\d+x?' '(?m)^\d+' 'a+b|b+a' '(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]' '\bGET\b|\bPOST\b')
for v in product fsmlocc5 product fsmlocc5; do
  if [ $v = product ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$R/coregex_amd/variants/libcoregex_hip_$v.so; fi
  echo "== $v"; timeout 300 python scripts/time_patterns.py "${PATS[@]}" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_c19_fsml_occ_$v.txt | cut -c1-30,150-260
done
