cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r4g
mkdir -p $o
python -m pytest tests/test_gpu_train_step.py tests/test_gpu_full_size_widened.py tests/test_gpu_multirank.py tests/test_gpu_domain_surfaces.py -x -q 2>&1 | tail -12 > $o/train_tests.txt
cat $o/train_tests.txt
for wl in e8_train e7_train; do
python bench.py --workload $wl --no_cpu_baseline --no_extras > $o/bench_$wl.json 2> $o/bench_$wl.err
python bench.py --workload $wl --joint --no_cpu_baseline --no_extras > $o/bench_${wl}_joint.json 2> $o/bench_${wl}_joint.err
done
python - <<'P'
import json
for n in ("e8_train","e8_train_joint","e7_train","e7_train_joint"):
    try:
        d=json.loads(open("gpurun_out/r4g/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["roofline"].get("backward"))
    except Exception as e:
        print(n, "failed", e)
P
