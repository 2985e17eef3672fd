cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
export CXG_LIB_PATH=$GRAFT_REPO_ROOT/coregex_amd/libcxg_prof.so
for PAT in '\berror\b' '\d+\.\d+x?' 'a+b|b+a' '(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'; do CXG_PROF=1 timeout 200 python scripts/time_patterns.py "$PAT" 2>&1 | grep -E "CXG_PROF|kernel_ms" | tail -2 | cut -c1-300; done
