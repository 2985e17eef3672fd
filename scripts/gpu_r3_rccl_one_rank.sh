cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
echo "--- torchrun, 1 rank, RCCL"; timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-pmc 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-700
echo "--- bench.py --gpus 2 on a 1-GPU box"; timeout 120 python bench.py --gpus 2 --steps 3 --warmup 1 2>&1 | tail -2 | cut -c1-300
