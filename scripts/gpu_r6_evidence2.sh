# Round 6, second evidence run (the tree with the pair kernel): config 3 (bench line with live PMC traffic, rocprofv3 kernel stats), the default
# line (north star), smoke, the GPU tier, device fuzz modes that draw literal sets.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for N in 3; do
  timeout 300 python $R/bench.py --config $N --steps 20 --warmup 5 --no-north-star > $R/gpurun_out/r06_cfg${N}_bench.json 2> $R/gpurun_out/r06_cfg${N}_bench.err; echo "cfg $N bench rc=$?"
  rm -rf /tmp/prof_$N
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$N -o cfg$N -- python $R/bench.py --config $N --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async > /tmp/prof_$N.log 2>&1; echo "cfg $N stats rc=$?"
  db=$(find /tmp/prof_$N -name "*.db" | head -1)
  [ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $R/gpurun_out/r06_cfg${N}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config $N --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-north-star --no-async" > /dev/null
done
CXG_NO_TEDDY_PAIR=1 timeout 300 python $R/bench.py --config 3 --steps 20 --warmup 5 --no-north-star --no-cpu-baseline --no-async > $R/gpurun_out/r06_cfg3_wave_kernel_bench.json 2> /dev/null; echo "cfg 3 wave kernel rc=$?"
cd $R
python - <<'PY'
import json
for f in ("gpurun_out/r06_cfg3_bench.json", "gpurun_out/r06_cfg3_wave_kernel_bench.json"):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1]); r = b["roofline"]
        print(f.split("/")[-1], "value", b["value"], "ms/step", b["ms_per_step"], r["kernel"], r["kernel_ms_avg"], "frac", r["frac"], "traffic", r.get("traffic"), "alg", r["algorithmic_bytes_per_launch"])
    except Exception as e:
        print(f, "unreadable", e)
PY
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
export CXG_PAIR_MIN_BYTES=0   # (the fuzz haystacks are short: by default they would stay on the wave kernel)
FUZZ_FOLD=1 timeout 200 python scripts/gpu_fuzz.py 907 250 > gpurun_out/r06_e2_gpu_fuzz_fold.txt 2>&1; tail -1 gpurun_out/r06_e2_gpu_fuzz_fold.txt | cut -c1-300
timeout 300 python scripts/gpu_fuzz.py 901 300 > gpurun_out/r06_e2_gpu_fuzz_general.txt 2>&1; tail -1 gpurun_out/r06_e2_gpu_fuzz_general.txt | cut -c1-300
FUZZ_FEW=1 timeout 200 python scripts/gpu_fuzz.py 908 80 > gpurun_out/r06_e2_gpu_fuzz_few.txt 2>&1; tail -1 gpurun_out/r06_e2_gpu_fuzz_few.txt | cut -c1-300
unset CXG_PAIR_MIN_BYTES
timeout 300 python scripts/gpu_stress_pair.py 20000 > gpurun_out/r06_e2_stress_pair.txt 2>&1; tail -1 gpurun_out/r06_e2_stress_pair.txt
timeout 120 python scripts/pair_sizes_big.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_e2_pair_sizes.txt
CXG_NO_TEDDY_PAIR=1 CXG_PAIR_TAPER=wave timeout 120 python scripts/pair_sizes_big.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_e2_pair_sizes.txt
timeout 1800 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r06_e2_pytest_gpu.log 2>&1; echo pytest=$?; tail -9 gpurun_out/r06_e2_pytest_gpu.log | cut -c1-300
cd /tmp; timeout 500 python $R/bench.py > $R/gpurun_out/r06_e2_bench_default.json 2> $R/gpurun_out/r06_e2_bench_default.err; echo "default bench rc=$?"; tail -c 600 $R/gpurun_out/r06_e2_bench_default.json
