#!/bin/bash
# Where k_diversity's time goes (on the GPU box): builds diversity_kernels.hip with sections compiled out (PSTL_DIV_SKIP bit mask,
# timing only) and times each build with tools/dbg/div_time.py.   DIV_VARIANTS="1 2 32" tools/dbg/div_ablation.sh
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; c=$root/pstl_diffusion_policy_amd/csrc; out=/tmp/pv; mkdir -p $out
cd $root; python3 tools/dbg/div_time.py 2>/dev/null | tail -1
for v in ${DIV_VARIANTS:-1 2 32 8 16 31}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPSTL_DIV_SKIP=$v -c $c/diversity_kernels.hip -o $out/d$v.o || continue
  hipcc --offload-arch=gfx950 -shared -fPIC $c/stl_kernels.o $c/mlp_kernels.o $c/train_kernels.o $out/d$v.o $c/stl_program.o -o $out/lib$v.so
  echo -n "skip=$v: "; python3 tools/dbg/with_lib.py $out/lib$v.so tools/dbg/div_time.py 2>/dev/null | tail -1 | cut -c1-50
done
