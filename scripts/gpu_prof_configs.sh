# rocprofv3 kernel stats for the five BASELINE configs (1 GiB each) -> gpurun_out/prof_cfgs/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/prof_cfgs
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$GRAFT_REPO_ROOT timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfgs -o cfgs -- python $GRAFT_REPO_ROOT/scripts/time_configs.py > $GRAFT_REPO_ROOT/gpurun_out/prof_cfgs.log 2>&1; echo prof=$?
grep cfg $GRAFT_REPO_ROOT/gpurun_out/prof_cfgs.log
