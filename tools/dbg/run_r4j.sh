cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r4j
mkdir -p $o
python bench.py > $o/bench_default.json 2> $o/bench.err
python3 - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/r4j/bench_default.json').read().strip().splitlines()[-1])
print('ms',d['ms_per_step'],'whole',d['roofline']['whole_step_frac'], 'frac', d['roofline']['frac'])
for k,v in d['also'].items(): print(k, v['ms_per_step'], v['value'], v['roofline']['frac'], v.get('backward',{}).get('ms'), v.get('backward',{}).get('achieved_GBps'))
for s in d['sweep']: print(s)
print(d['cpu_baseline']['sample'], d['cpu_baseline']['value'], d['cpu_baseline']['gpu_same_inputs'])
P
