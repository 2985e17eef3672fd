# A/B of compiler scheduling strategies for the headline kernel (libs built with -mllvm flags into coregex_amd/libcxg_<v>.so)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do for v in base ilp mem nomisched; do
  if [ $v = base ]; then unset CXG_LIB_PATH; else export CXG_LIB_PATH=$GRAFT_REPO_ROOT/coregex_amd/libcxg_$v.so; fi
  timeout 200 python bench.py --config 2 --steps 40 --warmup 5 --no-cpu-baseline --no-pmc | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'cfg2', d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
done; done
