set -x
cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r4c
mkdir -p $o
python tools/sweep_sizes.py --rollout_only --steps 100 --scenes 64,80,96,112,128,144,160,176,192,208,256,512,1024,4096 > $o/chain_us_per_tile_step.txt 2>&1
python tools/sweep_sizes.py --detail --diversity --scenes 1,16,32,128 > $o/sweep_with_diversity.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for sc in 16 128; do
rocprofv3 --kernel-trace --stats --output-format csv -d $o/st_$sc -o run -- python3 $GRAFT_REPO_ROOT/tools/sweep_sizes.py --diversity --scenes $sc --reps 20 > $o/sweep_rocprof_$sc.txt 2>&1
find $o/st_$sc -name "*kernel_stats.csv" -exec cp {} $o/kernel_stats_scenes$sc.csv \;
rm -rf $o/st_$sc
done
cat $o/chain_us_per_tile_step.txt $o/sweep_with_diversity.txt
for sc in 16 128; do head -25 $o/kernel_stats_scenes$sc.csv | cut -d, -f1-4 | cut -c1-150; done
