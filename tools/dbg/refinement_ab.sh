#!/bin/bash
# --refinement with and without the per-scene packing of the rows to mix (on the GPU box): times both builds at full size
# and checks that their refined controls are bit-identical.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; c=$root/pstl_diffusion_policy_amd/csrc; out=/tmp/pv; mkdir -p $out
cd $root
python3 tools/dbg/refinement_time.py /tmp/new.pt 2>/dev/null | tail -1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Xarch_device -mllvm=-misched=gcn-iterative-ilp -DPSTL_MIX_NO_COMPACT -c $c/stl_kernels.hip -o $out/s.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC $out/s.o $c/mlp_kernels.o $c/train_kernels.o $c/diversity_kernels.o $c/stl_program.o -o $out/libold.so
python3 tools/dbg/with_lib.py $out/libold.so tools/dbg/refinement_time.py /tmp/old.pt 2>/dev/null | tail -1
python3 -c "
import torch; a=torch.load('/tmp/new.pt'); b=torch.load('/tmp/old.pt'); print('identical:', torch.equal(a,b), 'max diff', float((a-b).abs().max()))"
timeout 600 python3 -m pytest tests/test_gpu_refinement.py -x -q 2>&1 | tail -2
