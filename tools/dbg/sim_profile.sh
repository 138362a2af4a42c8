#!/bin/bash
# kernel time vs wall latency of the closed-loop caller (on the GPU box): tools/dbg/sim_profile.sh
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export PYTHONPATH=$root
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st_sim
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_sim -o run -- python3 -m pstl_diffusion_policy_amd.nusc_sim --diffusion --load_stlp --rect_head --flex --diverse_loss --multi_cands 5 --guidance --guidance_before 10 --guidance_niters 1 --guidance_lr 0.04 --n_neighbors ${SIM_K:-8} --n_randoms 64 --diffusion_steps 100 --n_trials 12 --allow_random_init 2>&1 | grep -i "median\|sim 1" | tail -4
f=$(find /tmp/st_sim -name "*kernel_stats.csv")
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.2f = %.3f per simulation step (12 steps)" % (tot / 1e6, tot / 1e6 / 12))
for r in rows[:26]:
    print("%-70s calls %5s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
P
