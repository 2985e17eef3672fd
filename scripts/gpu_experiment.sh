# Timing experiments for the digit kernel (phase cycle counters + debug switches).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "CXG_PROF=1" "CXG_DEBUG=1" "CXG_DEBUG=2" "CXG_DEBUG=3"; do
  echo "== $v"; env $v timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep -E 'CXG_PROF|kernel_ms' | tail -2 | cut -c1-400
done
