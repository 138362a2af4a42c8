cd $GRAFT_REPO_ROOT
for i in 1 2; do
for j in "" "--joint"; do
python bench.py --workload e8_train $j --no_cpu_baseline --no_extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('check   ', '$j', round(d['ms_per_step'],2), round(d['roofline']['kernel_ms'],2))"
PSTL_DBG_NO_TRAIN_CHECK=1 python bench.py --workload e8_train $j --no_cpu_baseline --no_extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no check', '$j', round(d['ms_per_step'],2), round(d['roofline']['kernel_ms'],2))"
done; done
nproc; cat /proc/loadavg
