#!/bin/bash
# As build_variants.sh, for chain2_kernels.hip (k_chain2):  tools/dbg/build_variants2.sh name1:"-DPSTL_C2_ABL=1" ...
# The other translation units are the in-tree objects (run `python -m pstl_diffusion_policy_amd.build` first).
root=$(cd "$(dirname "$0")/../.." && pwd)
c=$root/pstl_diffusion_policy_amd/csrc
out=$root/tools/dbg/_variants
mkdir -p $out
pids=()
for v in "$@"; do
  n=${v%%:*}; flags=${v#*:}
  (
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xarch_device -mllvm=-amdgpu-mfma-vgpr-form $flags \
      -c $c/chain2_kernels.hip -o $out/c2_$n.o 2> $out/build_$n.log \
    && hipcc --offload-arch=gfx950 -shared -fPIC $c/stl_kernels.o $c/mlp_kernels.o $c/train_kernels.o $out/c2_$n.o $c/diversity_kernels.o \
         $c/stl_program.o -o $out/libpstl_$n.so && rm -f $out/c2_$n.o && echo "built $n" || { echo "FAILED $n"; tail -5 $out/build_$n.log; }
  ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
ls -la $out/*.so 2>/dev/null
