#!/bin/bash
# Regenerates every measured artefact of a round on the GPU box (run through gpurun):
#   tools/refresh_profiles.sh r2      -> gpurun_out/profiles_r2/*   (copy into profiles/r2/ afterwards)
# bench lines of every workload, rocprofv3 kernel stats/trace of the default bench, three separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ/GRBM) and their summary (tools/summarize_pmc.py).
set -u
r=${1:-r6}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/profiles_$r
mkdir -p "$out"
cd "$root"
b() { name=$1; shift; python3 bench.py "$@" 2> "$out/$name.err" | tail -1 > "$out/$name.json"; cut -c1-230 "$out/$name.json"; }
b bench_default
b bench_fp32_mfma --chain_waves 8 --no_cpu_baseline
b bench_split_bf16 --chain_waves 32 --no_cpu_baseline
b bench_e5 --workload e5 --no_cpu_baseline
b bench_e7 --workload e7 --no_cpu_baseline
b bench_k8_s100 --neighbors 8 --diffusion_steps 100 --no_cpu_baseline
b bench_e8_train --workload e8_train --no_cpu_baseline
b bench_e7_train --workload e7_train --no_cpu_baseline
b bench_e8_train_joint --workload e8_train --joint --no_cpu_baseline
b bench_e7_train_joint --workload e7_train --joint --no_cpu_baseline
b bench_trajopt --workload trajopt --steps 3 --warmup 1
b bench_big_shard --scenes 32768 --steps 3 --warmup 1 --no_cpu_baseline
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 "$root/bench.py" --no_cpu_baseline --no_extras > "$out/bench_default_under_rocprof.json" 2> "$out/stats.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_fp32" -o run -- python3 "$root/bench.py" --no_cpu_baseline --no_extras --chain_waves 8 --steps 5 > "$out/bench_fp32_under_rocprof.json" 2> "$out/stats_fp32.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_train" -o run -- python3 "$root/bench.py" --no_cpu_baseline --no_extras --workload e8_train --steps 5 > "$out/bench_e8_train_under_rocprof.json" 2> "$out/stats_train.err"
# (a fourth pass, "VALU": the vector-issue counters that show the one-row-per-lane STL kernels to be VALU-issue-bound)
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1 | sed 's/SQ_WAVE_CYCLES/SQ/; s/SQ_INSTS_VALU/VALU/')
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$out/pmc_$tag" -o run -- python3 "$root/bench.py" --no_cpu_baseline --no_extras --steps 2 --warmup 1 > /dev/null 2> "$out/pmc_$tag.err"
done
cd "$root"
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/bench_default_kernel_stats.csv" \;
find "$out/stats_fp32" -name "*kernel_stats.csv" -exec cp {} "$out/bench_fp32_mfma_kernel_stats.csv" \;
find "$out/stats_train" -name "*kernel_stats.csv" -exec cp {} "$out/bench_e8_train_kernel_stats.csv" \;
tail -1 "$out/bench_fp32_under_rocprof.json" > "$out/tmp.json" && mv "$out/tmp.json" "$out/bench_fp32_under_rocprof.json"
tail -1 "$out/bench_e8_train_under_rocprof.json" > "$out/tmp.json" && mv "$out/tmp.json" "$out/bench_e8_train_under_rocprof.json"
find "$out/stats" -name "*kernel_trace.csv" -exec cp {} "$out/bench_default_kernel_trace.csv" \;
for tag in FETCH_SIZE WRITE_SIZE SQ VALU; do
  find "$out/pmc_$tag" -name "*counter_collection.csv" -exec cp {} "$out/pmc_${tag}_counter_collection.csv" \;
done
tail -1 "$out/bench_default_under_rocprof.json" > "$out/tmp.json" && mv "$out/tmp.json" "$out/bench_default_under_rocprof.json"
python3 tools/summarize_pmc.py "$out" 786432 2 > "$out/pmc_summary.json"     # rows and neighbours of the default bench
rm -rf "$out/stats" "$out/stats_fp32" "$out/stats_train" "$out"/pmc_FETCH_SIZE "$out"/pmc_WRITE_SIZE "$out"/pmc_SQ "$out"/pmc_VALU
python3 tools/dbg/chain_time.py 0 116 0:nonoise 32 8 > "$out/chain_ablation_times.txt" 2>/dev/null
head -c 900 "$out/pmc_summary.json"; echo; head -8 "$out/bench_default_kernel_stats.csv" | cut -c1-170
