cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r4m
mkdir -p $o
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $o/all_gpu_tests.txt
python bench.py --no_cpu_baseline --no_extras > $o/bench_default_noextras.json 2> $o/bench.err
python tools/closed_loop_latency.py > $o/closed_loop_latency.txt 2>&1
cat $o/all_gpu_tests.txt $o/closed_loop_latency.txt
python3 - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/r4m/bench_default_noextras.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['whole_step_frac'])
for k,v in d['roofline']['stl_kernels'].items(): print(k, v['ms_per_step'])
P
