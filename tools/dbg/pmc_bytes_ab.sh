#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of k_guidance_iter per launch for the in-tree library and for variants (tools/dbg/_variants/libpstl_<name>.so):
#   tools/dbg/pmc_bytes_ab.sh base oldmix ...        (GPU box; separate --pmc passes, kernel trace only)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for n in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    out=$root/gpurun_out/pmc_ab/${n}_$c; rm -rf $out; mkdir -p $out
    cd /tmp && export TMPDIR=/tmp
    if [ "$n" = base ]; then
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o run -- python3 $root/bench.py --no_cpu_baseline --no_extras --steps 2 --warmup 1 > /dev/null 2> $out/err.txt
    else
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o run -- python3 $root/tools/dbg/with_lib.py $root/tools/dbg/_variants/libpstl_$n.so bench.py --no_cpu_baseline --no_extras --steps 2 --warmup 1 > /dev/null 2> $out/err.txt
    fi
    cd $root
    python3 - "$out" "$n" "$c" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "k_guidance_iter" in r["Kernel_Name"]] if f else []
print("%-10s %-10s k_guidance_iter: %.1f MB per launch over %d launches" % (sys.argv[2], sys.argv[3], sum(v) / max(len(v), 1) / 1024.0, len(v)))
PY
  done
done
