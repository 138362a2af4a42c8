set -x
cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r4b
mkdir -p $o
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $o/all_gpu_tests.txt
python bench.py > $o/bench_default.json 2> $o/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $o/pmc_VALU -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --no_extras --steps 2 --warmup 1 > /dev/null 2> $o/pmc_VALU.err
find $o/pmc_VALU -name "*counter_collection.csv" -exec cp {} $o/pmc_VALU_counter_collection.csv \;
rm -rf $o/pmc_VALU
cd $GRAFT_REPO_ROOT
python3 tools/summarize_pmc.py $o 786432 2 > $o/pmc_summary_valu_only.json
cat $o/all_gpu_tests.txt; cut -c1-1500 $o/bench_default.json
