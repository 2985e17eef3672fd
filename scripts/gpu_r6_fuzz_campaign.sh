# Round 6: device fuzz campaign on the final tree (every mode of scripts/gpu_fuzz.py, fresh seeds).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" timeout 420 python scripts/gpu_fuzz.py $SEED $N > gpurun_out/r06_campaign_$name.txt 2>&1; echo "$name: $(grep '^seed' gpurun_out/r06_campaign_$name.txt | cut -c1-260) mismatches $(grep -c MISMATCH gpurun_out/r06_campaign_$name.txt)"; }
SEED=801 N=350 run general1 X=1
SEED=802 N=350 run general2 X=1
SEED=803 N=400 run look FUZZ_LOOK=1
SEED=804 N=300 run end FUZZ_END=1
SEED=805 N=300 run text FUZZ_TEXT=1
SEED=806 N=250 run wide FUZZ_WIDE=1
SEED=807 N=250 run fold FUZZ_FOLD=1
SEED=808 N=80 run few FUZZ_FEW=1
