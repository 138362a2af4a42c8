#!/bin/bash
# tools/dbg/variant_run.sh "<extra hipcc flags for mlp_kernels.hip>" script.py args...   (on the GPU box)
root=$(cd "$(dirname "$0")/../.." && pwd)
c=$root/pstl_diffusion_policy_amd/csrc
out=/tmp/pstl_variants; mkdir -p $out
flags=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xarch_device -mllvm=-misched-prera-direction=topdown $flags -c $c/mlp_kernels.hip -o $out/vr.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC $c/stl_kernels.o $out/vr.o $c/train_kernels.o $c/chain2_kernels.o $c/chain2_kernels_p1.o $c/chain2_kernels_p2.o $c/diversity_kernels.o $c/stl_program.o -o $out/libpstl_vr.so
cd $root && python3 tools/dbg/with_lib.py $out/libpstl_vr.so "$@"
