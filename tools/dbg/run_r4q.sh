cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -3
bash tools/refresh_profiles.sh r4 > gpurun_out/refresh_r4.log 2>&1
tail -5 gpurun_out/refresh_r4.log | cut -c1-300
o=$GRAFT_REPO_ROOT/gpurun_out/profiles_r4
python tools/closed_loop_latency.py > $o/closed_loop_latency_graph_vs_eager.json 2>/dev/null
python tools/sweep_sizes.py --detail --diversity --scenes 1,16,128,512 > $o/sweep_e7_guid_detail.txt 2>/dev/null
python tools/sweep_sizes.py --detail --workload e7 --steps 100 --scenes 128 > $o/sweep_e7_100steps_24576rows.txt 2>/dev/null
cat $o/closed_loop_latency_graph_vs_eager.json $o/sweep_e7_100steps_24576rows.txt
