#!/bin/bash
# Per-kernel time of one bench.py invocation on the GPU box (run through gpurun):
#   tools/prof_stats.sh <tag> [bench.py args...]      -> gpurun_out/prof_<tag>/<tag>_kernel_stats.csv (+ trace)
# rocprofv3 gets the program itself after `--` (no shell / env hop: see the round notes on exec after GPU init).
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$tag" -- python3 "$root/bench.py" --no_cpu_baseline "$@" > "$out/run.log" 2>&1
find "$out" -name "*kernel_stats.csv" | head -1 | xargs -r head -${PROF_HEAD:-14}
