#!/bin/bash
# usage: run_libs.sh <name...>  -- bench with tools/dbg/_variants/libpstl_<name>.so, prints step / STL kernel times
for n in "$@"; do
  PSTL_HIP_LIB=tools/dbg/_variants/libpstl_$n.so timeout 200 python bench.py --steps 5 --warmup 2 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['stl_kernels']; print('$n: step %.2f ms, chain %.2f, guidance %.3f, score %.3f, sat %.6f' % (d['ms_per_step'], d['roofline']['kernel_ms'], r['guidance']['ms_per_step'], r['score']['ms_per_step'], d['stl_sat_rate']))"
done
