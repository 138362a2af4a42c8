set -x
cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r4d
mkdir -p $o
python -m pytest tests/test_gpu_closed_loop.py tests/test_gpu_domain_surfaces.py -x -q -s 2>&1 | tail -15 > $o/closed_loop_tests.txt
python tools/closed_loop_latency.py > $o/closed_loop_latency.txt 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $o/all_gpu_tests.txt
python bench.py --no_cpu_baseline --no_extras > $o/bench_default_noextras.json 2> $o/bench.err
python tools/sweep_sizes.py --rollout_only --steps 100 --scenes 112,128,144,267 > $o/chain_balanced.txt 2>&1
cat $o/closed_loop_tests.txt $o/closed_loop_latency.txt $o/all_gpu_tests.txt $o/chain_balanced.txt; cut -c1-400 $o/bench_default_noextras.json
