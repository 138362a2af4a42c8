cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r4p
mkdir -p $o
python -m pytest tests/test_gpu_train_step.py tests/test_gpu_full_size_widened.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -3
for j in "" "--joint"; do
python bench.py --workload e8_train $j --no_cpu_baseline --no_extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('e8 $j', round(d['ms_per_step'],2), round(d['roofline']['kernel_ms'],2), d['roofline']['backward']['ms'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/st -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --no_extras --workload e8_train --steps 5 > /dev/null 2>&1
find $o/st -name "*kernel_stats.csv" -exec cp {} $o/bench_e8_train_kernel_stats.csv \;
rm -rf $o/st
python3 - <<'P'
import csv,re
rows=list(csv.DictReader(open("/root/repo/gpurun_out/r4p/bench_e8_train_kernel_stats.csv")))
for r in rows[:12]:
    n=r['Name']; m=re.search(r'(k_\w+(<[^>]*>)?)',n); nm=m.group(1) if m else n[:60]
    print('%-50s calls %4s avg %9.1f us'%(nm[:50], r['Calls'], float(r['AverageNs'])/1e3))
P
