cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
for v in base cap5 cap4; do
  if [ $v = base ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$GRAFT_REPO_ROOT/coregex_amd/libcxg_$v.so; fi
  echo "== $v"
  timeout 300 python bench.py --config 5 --steps 30 --warmup 5 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg5', d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
  timeout 300 python scripts/time_patterns.py '[A-Z][a-z]+' '(\w+)=(\d+)' '\w+@\w+' '(\d+)\.(\d+)\.(\d+)\.(\d+)' 2>&1 | grep kernel_ms | cut -c1-150
done
