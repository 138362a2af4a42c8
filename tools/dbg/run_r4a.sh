set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_domain_surfaces.py -x -q 2>&1 | tail -30 > gpurun_out/r4a/domain_tests.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4a/all_gpu_tests.txt
python tools/sweep_sizes.py --detail > gpurun_out/r4a/sweep_e7_guid.txt 2>&1
python tools/sweep_sizes.py --detail --workload e7 --steps 100 --scenes 128 > gpurun_out/r4a/sweep_e7_100.txt 2>&1
python bench.py > gpurun_out/r4a/bench_default.json 2> gpurun_out/r4a/bench_default.err
python bench.py --workload e8_train --no_cpu_baseline > gpurun_out/r4a/bench_e8_train.json 2> gpurun_out/r4a/bench_e8_train.err
cat gpurun_out/r4a/domain_tests.txt gpurun_out/r4a/all_gpu_tests.txt gpurun_out/r4a/sweep_e7_guid.txt gpurun_out/r4a/sweep_e7_100.txt
