# Round 6: second device fuzz campaign, on the final tree with the pair kernel serving literal sets of every length (fresh seeds 821-828; the claim scheme of the final tree: one counter, static first groups).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; export CXG_PAIR_MIN_BYTES=0   # literal sets on the pair kernel at every length
run() { name=$1; shift; env "$@" timeout 420 python scripts/gpu_fuzz.py $SEED $N > gpurun_out/r06_campaign3_$name.txt 2>&1; echo "$name: $(grep '^seed' gpurun_out/r06_campaign3_$name.txt | cut -c1-260) mismatches $(grep -c MISMATCH gpurun_out/r06_campaign3_$name.txt)"; }
SEED=821 N=350 run general1 X=1
SEED=822 N=350 run general2 X=1
SEED=823 N=400 run look FUZZ_LOOK=1
SEED=824 N=300 run end FUZZ_END=1
SEED=825 N=300 run text FUZZ_TEXT=1
SEED=826 N=250 run wide FUZZ_WIDE=1
SEED=827 N=250 run fold FUZZ_FOLD=1
SEED=828 N=80 run few FUZZ_FEW=1
