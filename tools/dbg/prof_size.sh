#!/bin/bash
# Per-kernel time of one bench step at a given batch size (eager launches, so that every kernel shows up by name):
#   tools/dbg/prof_size.sh 128      (scenes; 128 = the paper's 24 576 rows, 512 = 98 304 rows)   -> gpurun_out/prof_size_<scenes>/
root=${GRAFT_REPO_ROOT:-/root/repo}; sc=${1:-128}
out=$root/gpurun_out/prof_size_$sc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -o run -- python3 $root/bench.py --scenes $sc --no_cpu_baseline --no_extras --steps 20 --no_graph > $out/bench.json 2> $out/err.txt
cd $root
f=$(find $out/st -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
n=23
for r in rows[:14]:
    print("%-84s /step %5.1f avg %7.1f us  per step %7.1f us %5.1f%%"%(r["Name"][:84], int(r["Calls"])/n, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/n/1e3, float(r["Percentage"])))
print("sum per step %.1f us"%(tot/n/1e3))
PY
tail -1 $out/bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
