#!/bin/bash
# Interleaved A/B of whole-library variants on the GPU box: tools/dbg/ab_libs.sh "<bench args>" base name1 name2 ...  (3 rounds)
# "base" = the in-tree library, nameN = tools/dbg/_variants/libpstl_<nameN>.so
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $root
bargs=$1; shift
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['stl_kernels']; print('$1: step %.2f ms, chain %.3f ms (frac %.3f), guidance %.3f, score %.3f, sat %.6f' % (d['ms_per_step'], r['kernel_ms'], r['frac'], s.get('guidance',{}).get('ms_per_step',0), s.get('score',{}).get('ms_per_step',0), d['stl_sat_rate']))"; }
for round in 1 2 3; do
  for n in "$@"; do
    if [ "$n" = base ]; then python3 bench.py --no_cpu_baseline --no_extras $bargs 2>/dev/null | tail -1 | line base
    else python3 tools/dbg/with_lib.py tools/dbg/_variants/libpstl_$n.so bench.py --no_cpu_baseline --no_extras $bargs 2>/dev/null | tail -1 | line $n; fi
  done
done
