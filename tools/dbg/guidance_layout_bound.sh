#!/bin/bash
# What another layout of the guided phase's state (mu, row-major (N,40): a wavefront's rows are 480 bytes apart) could gain in
# k_guidance_iter at most -- two timing-only builds of stl_kernels.hip (results are garbage), built HERE, timed on the GPU box:
#   nostore:  -DPSTL_G_ABL=1  no state / candidate stores at all
#   elemmajor: -DPSTL_G_ABL=2  every read and write of the state at element-major addresses (lanes 12 bytes apart)
#   tools/dbg/guidance_layout_bound.sh build      (no GPU needed)
#   tools/dbg/guidance_layout_bound.sh time       (on the GPU box: per-launch time at 4096 scenes = 786 432 rows)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; c=$root/pstl_diffusion_policy_amd/csrc; out=$root/tools/dbg/_variants; mkdir -p $out
if [ "$1" = build ]; then
  for v in nostore:1 elemmajor:2; do
    n=${v%%:*}; f=${v#*:}
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Xarch_device -mllvm=-misched=gcn-iterative-ilp -DPSTL_G_ABL=$f \
      -c $c/stl_kernels.hip -o $out/stl_$n.o && \
    hipcc --offload-arch=gfx950 -shared -fPIC $out/stl_$n.o $c/mlp_kernels.o $c/train_kernels.o $c/chain2_kernels.o $c/chain2_kernels_p1.o $c/chain2_kernels_p2.o $c/diversity_kernels.o \
      $c/stl_program.o -o $out/libpstl_g_$n.so && rm -f $out/stl_$n.o && echo built $n &
  done; wait
else
  cd $root
  echo "in-tree kernel:"; SIZES=4096 python3 tools/dbg/guidance_by_size.py 2>/dev/null
  for n in nostore elemmajor; do echo "$n:"; SIZES=4096 python3 tools/dbg/with_lib.py $out/libpstl_g_$n.so tools/dbg/guidance_by_size.py 2>/dev/null; done
fi
