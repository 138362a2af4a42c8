#!/bin/bash
# PMC counters of one bench.py invocation (own run, kernel-trace only -- never combined with sys/hip traces):
#   tools/prof_pmc.sh <tag> "<COUNTER ...>" [bench.py args...]  -> gpurun_out/pmc_<tag>/*counter_collection.csv
set -u
tag=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out" -o "$tag" -- python3 "$root/bench.py" --no_cpu_baseline "$@" > "$out/run.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter csv"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][-60:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-28s %.4g  (per launch, %d launches)" % (c, v / cnt[(k, c)], cnt[(k, c)]))
PY
