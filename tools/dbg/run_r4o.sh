cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r4o
mkdir -p $o
python -m pytest tests -m gpu -q 2>&1 | tail -4 > $o/all_gpu_tests.txt
python bench.py --no_cpu_baseline --no_extras > $o/bench_default_noextras.json 2> $o/bench.err
python bench.py --workload trajopt --steps 3 --warmup 1 > $o/bench_trajopt.json 2>> $o/bench.err
python bench.py --workload e8_train --no_cpu_baseline --no_extras > $o/bench_e8.json 2>> $o/bench.err
cat $o/all_gpu_tests.txt
python3 - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/r4o/bench_default_noextras.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['whole_step_frac'])
for k,v in d['roofline']['stl_kernels'].items(): print(k, v['ms_per_step'])
t=json.loads(open('/root/repo/gpurun_out/r4o/bench_trajopt.json').read().strip().splitlines()[-1]); print('trajopt', t['ms_per_step'], t['value'])
t=json.loads(open('/root/repo/gpurun_out/r4o/bench_e8.json').read().strip().splitlines()[-1]); print('e8', t['ms_per_step'], t['roofline']['kernel_ms'], t['roofline']['backward']['ms'])
P
