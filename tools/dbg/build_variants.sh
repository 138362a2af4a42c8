#!/bin/bash
# Builds libpstl_hip variants with -DPSTL_EXP=<n> (timing experiments inside mlp_kernels.hip) into tools/dbg/_variants/
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
c=$root/pstl_diffusion_policy_amd/csrc
out=$root/tools/dbg/_variants
mkdir -p "$out"
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPSTL_EXP=$n -c $c/mlp_kernels.hip -o $out/mlp_$n.o
  hipcc --offload-arch=gfx950 -shared -fPIC $c/stl_kernels.o $out/mlp_$n.o $c/train_kernels.o $c/diversity_kernels.o $c/stl_program.o -lrocblas -o $out/libpstl_$n.so
  rm $out/mlp_$n.o
done
