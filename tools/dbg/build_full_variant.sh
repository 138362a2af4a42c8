#!/bin/bash
# A variant of the WHOLE library with extra compiler flags (timing A/B of a change that touches several units), built HERE into
# tools/dbg/_variants/libpstl_<name>.so with the per-unit flags of pstl_diffusion_policy_amd/build.py:
#   tools/dbg/build_full_variant.sh philox10 -DPSTL_PHILOX_ROUNDS=10
# then on the GPU box:  python3 tools/dbg/with_lib.py tools/dbg/_variants/libpstl_philox10.so bench.py --no_cpu_baseline --no_extras
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/tools/dbg/_variants; mkdir -p $out/obj_$1
name=$1; shift
cd $root && python3 - "$name" "$@" <<'PY'
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
from pstl_diffusion_policy_amd.build import UNITS, CSRC
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join("tools/dbg/_variants", "obj_" + name)
procs, objs = [], []
for u in UNITS:
    o = os.path.join(out, (u[2] if len(u) > 2 else u[0].replace(".hip", ".o")))
    objs.append(o)
    procs.append(subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + u[1] + extra + ["-c", os.path.join(CSRC, u[0]), "-o", o]))
assert all(p.wait() == 0 for p in procs)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", "tools/dbg/_variants/libpstl_%s.so" % name])
subprocess.check_call(["rm", "-rf", out])
print("built tools/dbg/_variants/libpstl_%s.so" % name)
PY
