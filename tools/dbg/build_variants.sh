#!/bin/bash
# Builds variants of mlp_kernels.hip HERE (hipcc cross-compiles gfx950 without a GPU) into tools/dbg/_variants/ -- the
# directory is git-ignored but travels to the GPU box with the snapshot -- for tools/dbg/time_variants.py to time there:
#   tools/dbg/build_variants.sh name1:"-DPSTL_EXP_A" name2:"-DPSTL_EXP_A -DPSTL_EXP_B" ...
# The other translation units are the in-tree objects (run `python -m pstl_diffusion_policy_amd.build` first).
root=$(cd "$(dirname "$0")/../.." && pwd)
c=$root/pstl_diffusion_policy_amd/csrc
out=$root/tools/dbg/_variants
mkdir -p $out
pids=()
for v in "$@"; do
  n=${v%%:*}; flags=${v#*:}
  sched="-Xarch_device -mllvm=-misched-prera-direction=topdown"     # the product build's scheduling flag (build.py)
  case "$flags" in *NOSCHED*) sched=""; flags=${flags//NOSCHED/};; esac   # NOSCHED in the flags: build without it
  (
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $sched $flags \
      -c $c/mlp_kernels.hip -o $out/mlp_$n.o 2> $out/build_$n.log \
    && hipcc --offload-arch=gfx950 -shared -fPIC $c/stl_kernels.o $out/mlp_$n.o $c/train_kernels.o $c/chain2_kernels.o $c/chain2_kernels_p1.o $c/chain2_kernels_p2.o $c/diversity_kernels.o \
         $c/stl_program.o -o $out/libpstl_$n.so && rm -f $out/mlp_$n.o && echo "built $n" || { echo "FAILED $n"; tail -5 $out/build_$n.log; }
  ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
ls -la $out/*.so 2>/dev/null
