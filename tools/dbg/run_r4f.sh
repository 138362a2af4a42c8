cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r4f
mkdir -p $o
python -m pytest tests/test_gpu_closed_loop.py tests/test_gpu_domain_surfaces.py tests/test_gpu_reference_surface.py -x -q -s 2>&1 | tail -8 > $o/closed_loop_tests.txt
python tools/closed_loop_latency.py > $o/closed_loop_latency.txt 2>&1
cat $o/closed_loop_tests.txt $o/closed_loop_latency.txt
