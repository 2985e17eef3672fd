# Round 6: second device fuzz campaign, on the final tree with the pair kernel serving literal sets of every length (fresh seeds 811-818).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; export CXG_PAIR_MIN_BYTES=0   # literal sets on the pair kernel at every length
run() { name=$1; shift; env "$@" timeout 420 python scripts/gpu_fuzz.py $SEED $N > gpurun_out/r06_campaign2_$name.txt 2>&1; echo "$name: $(grep '^seed' gpurun_out/r06_campaign2_$name.txt | cut -c1-260) mismatches $(grep -c MISMATCH gpurun_out/r06_campaign2_$name.txt)"; }
SEED=811 N=350 run general1 X=1
SEED=812 N=350 run general2 X=1
SEED=813 N=400 run look FUZZ_LOOK=1
SEED=814 N=300 run end FUZZ_END=1
SEED=815 N=300 run text FUZZ_TEXT=1
SEED=816 N=250 run wide FUZZ_WIDE=1
SEED=817 N=250 run fold FUZZ_FOLD=1
SEED=818 N=80 run few FUZZ_FEW=1
