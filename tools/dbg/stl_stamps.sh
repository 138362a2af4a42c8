#!/bin/bash
# Section cycle stamps of the one-wave STL kernels (k_guidance_iter, k_stl_forward): a -DPSTL_STL_STAMP build of stl_kernels.hip
# (stl_core.hpp, PSTL_ST), built HERE, run on the GPU box:
#   tools/dbg/stl_stamps.sh build         (no GPU needed)
#   tools/dbg/stl_stamps.sh run [--bs 4096] [--K 2]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; c=$root/pstl_diffusion_policy_amd/csrc; out=$root/tools/dbg/_variants; mkdir -p $out
if [ "$1" = build ]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Xarch_device -mllvm=-misched=gcn-iterative-ilp -DPSTL_STL_STAMP \
    -c $c/stl_kernels.hip -o $out/stl_stamp.o && \
  hipcc --offload-arch=gfx950 -shared -fPIC $out/stl_stamp.o $c/mlp_kernels.o $c/train_kernels.o $c/chain2_kernels.o $c/chain2_kernels_p1.o $c/chain2_kernels_p2.o \
    $c/diversity_kernels.o $c/stl_program.o $c/adam_kernels.o -o $out/libpstl_ststamp.so && rm -f $out/stl_stamp.o && echo built
else
  shift; cd $root; python3 tools/dbg/with_lib.py $out/libpstl_ststamp.so tools/dbg/stl_stamps.py "$@"
fi
