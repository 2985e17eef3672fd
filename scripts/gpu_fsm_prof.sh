# phase cycle counters of the transducer kernel (build with -DCXG_FSM_PROF=1 into a scratch copy of the library)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
for PAT in "$@"; do CXG_PROF=1 timeout 200 python scripts/time_patterns.py "$PAT" 2>&1 | grep -E "CXG_PROF|kernel_ms" | tail -2 | cut -c1-400; done
