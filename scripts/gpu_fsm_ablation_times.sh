cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
for v in base abl1 abl2 abl3 abl4; do
  if [ $v = base ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$GRAFT_REPO_ROOT/coregex_amd/libcxg_$v.so; fi
  echo "== $v (bit0 no entry walks, bit1 no lockstep walk, bit2 no rows/starts)"
  timeout 300 python scripts/time_patterns.py 'zq+x|x+zq' 'a+b|b+a' '\d+\.\d+x?' '\bzqerror\b' 2>&1 | grep kernel_ms | cut -c1-150
done
