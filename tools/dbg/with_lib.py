"""Runs a script of this repository against ANOTHER build of libpstl_hip.so (timing experiments only):
    python tools/dbg/with_lib.py tools/dbg/_variants/libpstl_x.so bench.py --no_cpu_baseline
The product never does this: ffi.lib() always loads the in-tree library; this runner re-points ffi.LIB_PATH before the
first call and says so on stderr."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pstl_diffusion_policy_amd import ffi  # noqa: E402

lib, script = os.path.abspath(sys.argv[1]), sys.argv[2]
assert os.path.exists(lib), lib
ffi.LIB_PATH = lib
print("with_lib: using %s" % lib, file=sys.stderr)
sys.argv = [script] + sys.argv[3:]
runpy.run_path(os.path.join(ROOT, script) if not os.path.isabs(script) else script, run_name="__main__")
