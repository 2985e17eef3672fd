cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do for v in base nobar; do
  if [ $v = base ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$GRAFT_REPO_ROOT/coregex_amd/libcxg_$v.so; fi
  echo "== $v"
  timeout 300 python scripts/time_patterns.py 'a+b|b+a' '\d+\.\d+x?' '\bzqerror\b' '\berror\b' '(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]' '(?m)^\d+' 2>&1 | grep kernel_ms | sed -E 's/ +Use[A-Za-z]+ +matches +[0-9]+//' | cut -c1-110
done; done
