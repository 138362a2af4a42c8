cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_closed_loop.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -5
bash tools/refresh_profiles.sh r4 > gpurun_out/refresh_r4.log 2>&1
tail -30 gpurun_out/refresh_r4.log
ls gpurun_out/profiles_r4 | head -80
