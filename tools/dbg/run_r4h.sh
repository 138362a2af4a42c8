cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r4h
mkdir -p $o
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/st -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --no_extras --workload e8_train --steps 5 > $o/bench_e8_train_under_rocprof.json 2> $o/stats_train.err
find $o/st -name "*kernel_stats.csv" -exec cp {} $o/bench_e8_train_kernel_stats.csv \;
rm -rf $o/st
python3 - <<'P'
import csv,re
rows=list(csv.DictReader(open("/root/repo/gpurun_out/r4h/bench_e8_train_kernel_stats.csv")))
for r in rows[:22]:
    n=r['Name']; m=re.search(r'(k_\w+(<[^>]*>)?)',n); nm=m.group(1) if m else n[:60]
    print('%-50s calls %4s avg %9.1f us'%(nm[:50], r['Calls'], float(r['AverageNs'])/1e3))
P
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $o/all_gpu_tests.txt
python bench.py --no_cpu_baseline --no_extras > $o/bench_default_noextras.json 2> $o/bench.err
cat $o/all_gpu_tests.txt
python3 - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/r4h/bench_default_noextras.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['whole_step_frac'])
for k,v in d['roofline']['stl_kernels'].items(): print(k, v['ms_per_step'])
P
