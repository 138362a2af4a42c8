#!/bin/bash
# usage: run_variants.sh <chain_waves> <n...>   -- times the chain kernel of each variant build
cw=$1; shift
for n in "$@"; do
  PSTL_HIP_LIB=tools/dbg/_variants/libpstl_$n.so timeout 200 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --chain_waves $cw 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exp $n cw $cw: step %.2f ms, chain %.3f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done
