cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r4n
mkdir -p $o
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $o/all_gpu_tests.txt
python tools/sweep_sizes.py --detail --scenes 1,16,64,128,256,512 > $o/sweep.txt 2>&1
python tools/closed_loop_latency.py > $o/closed_loop_latency.txt 2>&1
cat $o/all_gpu_tests.txt $o/sweep.txt $o/closed_loop_latency.txt
