#!/bin/bash
# How the STL kernels' time depends on the wavefronts resident per SIMD (is a kernel bound by instruction issue or by latency?):
# one library built with -DPSTL_DBG_LDS_ENV reads extra dynamic LDS from the environment, so fewer wavefronts fit a CU.
#   tools/dbg/build_full_variant.sh ldsenv -DPSTL_DBG_LDS_ENV      (here)
#   tools/dbg/occupancy_sweep.sh                                   (GPU box)
# k_stl_forward (selected formula, K = 2): 7.8 KB per wavefront = 5 per SIMD; pads 2200 / 4800 / 10500 B -> 4 / 3 / 2 per SIMD.
# k_guidance_iter: 12.9 KB (and 168 registers) = 3 per SIMD; pads 5500 / 20000 B -> 2 / 1 per SIMD.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $root
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['stl_kernels']; print('$1: step %.2f ms, guidance %.3f, score %.3f ms per step' % (d['ms_per_step'], s.get('guidance',{}).get('ms_per_step',0), s.get('score',{}).get('ms_per_step',0)))"; }
run() { python3 tools/dbg/with_lib.py tools/dbg/_variants/libpstl_ldsenv.so bench.py --no_cpu_baseline --no_extras --steps 6 2>/dev/null | tail -1 | line "$1"; }
for round in 1 2; do
  run "forward 5/SIMD, guidance 3/SIMD (as shipped)"
  PSTL_DBG_LDS_PAD_FORWARD=2200 run "forward 4/SIMD"
  PSTL_DBG_LDS_PAD_FORWARD=4800 run "forward 3/SIMD"
  PSTL_DBG_LDS_PAD_FORWARD=10500 run "forward 2/SIMD"
  PSTL_DBG_LDS_PAD_GUIDANCE=5500 run "guidance 2/SIMD"
  PSTL_DBG_LDS_PAD_GUIDANCE=20000 run "guidance 1/SIMD"
done
