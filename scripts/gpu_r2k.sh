cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
CFGS="5" bash scripts/gpu_r2_evidence.sh
cd $GRAFT_REPO_ROOT
FAT=$(python -c "print('|'.join(['word%02d'%i for i in range(20)]+['key%02dx'%i for i in range(12)]+['val%d'%i for i in range(10)]+['item','timeout','refused','denied','ordinal','keyword']))")
timeout 900 python scripts/time_patterns.py '\d+\.\d+\.\d+\.\d+' 'error' '\d+:\d+:\d+' '\d{4}-\d{2}-\d{2}' '[0-9a-f]{8}-[0-9a-f]{4}-[0-9a-f]{4}-[0-9a-f]{4}-[0-9a-f]{12}' '\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}' 'warning' 'error|warning|fatal|critical' "$FAT" '[\w]+' 'HTTP/\d\.\d' '(GET|POST|PUT) /[a-z/]+' '(\w+)@(\w+)\.(\w+)' '(a|ab)(c|bcd)' '(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)' '(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]' '\d+\.\d+x?' 'a+b|b+a' 'GET|POST /[a-z]+' '\berror\b' '\bGET\b' '\b\d+\.\d+\b' '(?m)^\d+' '(?m)^(GET|POST|PUT|DELETE|PATCH)' '(?m)[a-z]+$' > gpurun_out/r02_time_patterns.txt 2>&1
cat gpurun_out/r02_time_patterns.txt | cut -c1-200
timeout 600 python scripts/time_nosync.py > gpurun_out/r02_time_nosync.txt 2>&1; cat gpurun_out/r02_time_nosync.txt
