# round 2: bench lines of all five BASELINE configurations + rocprofv3 kernel stats of the headline command
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in 2 1 3 4 5; do timeout 400 python bench.py --config $c > gpurun_out/r02_cfg${c}_bench.json 2> gpurun_out/bench_cfg$c.err; echo cfg$c=$?; tail -c 1800 gpurun_out/r02_cfg${c}_bench.json; tail -3 gpurun_out/bench_cfg$c.err | grep -v amdgpu.ids; done
