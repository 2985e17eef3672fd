cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/stream scripts/microbench/stream.hip 2>&1 | tail -3
/tmp/stream | tee gpurun_out/microbench_stream.txt
