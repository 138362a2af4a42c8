#!/bin/bash
# Where a launch of the guidance kernel's latency layout spends its time (run ON the GPU box through gpurun): timing-only builds
# of stl_kernels.hip that leave the kernel before the geometry (9), after it (1), after the forward chains (2), the weights (3),
# the direct partials (4), the costate recursion (5) -- -DPSTL_DBG_GEXIT=n -- each timed per launch at 192 and 24 576 rows
# (tools/dbg/guidance_by_size.py), then the full kernel.    tools/dbg/guidance_stages.sh [K]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; c=$root/pstl_diffusion_policy_amd/csrc; out=/tmp/pstl_stages; mkdir -p $out
K=${1:-2}
for n in 9 1 2 3 4 5; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Xarch_device -mllvm=-misched=gcn-iterative-ilp -DPSTL_DBG_GEXIT=$n -c $c/stl_kernels.hip -o $out/s$n.o &
done; wait
cd $root
for n in 9 1 2 3 4 5; do
  hipcc --offload-arch=gfx950 -shared -fPIC $out/s$n.o $c/mlp_kernels.o $c/train_kernels.o $c/diversity_kernels.o $c/stl_program.o -o $out/lib$n.so
  echo "exit at stage $n:"; SIZES=1,128 python3 tools/dbg/with_lib.py $out/lib$n.so tools/dbg/guidance_by_size.py --K $K 2>/dev/null
done
echo "full kernel:"; SIZES=1,128 python3 tools/dbg/guidance_by_size.py --K $K 2>/dev/null
