"""Builds libpstl_hip.so (gfx950) in-tree with hipcc.  `python -m pstl_diffusion_policy_amd.build`."""
import os
import signal
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpstl_hip.so")
ARCH = "gfx950"

# (source, extra flags).  stl_kernels must not contract mul+add into fma: see csrc/stl_core.hpp.  Its device code is
# scheduled with the ILP-oriented machine scheduler: the default one serialises chains of packed-fp32 operations through
# one temporary and pays an s_nop between every two of them (a third of the issue slots of the circle-pair loop);
# results are bit-identical, the guidance kernel is 3 % faster.
UNITS = [("stl_kernels.hip", ["-ffp-contract=off", "-Xarch_device", "-mllvm=-misched=gcn-iterative-ilp"]),
         # mlp_kernels: top-down pre-RA list scheduling follows the hand-laid order of the chain kernel's fused block more
         # closely (multi-step launch -1.4 %, single-step -1.9 %, nothing else in the unit changes)
         ("mlp_kernels.hip", ["-Xarch_device", "-mllvm=-misched-prera-direction=topdown"]), ("train_kernels.hip", []),
         # chain2_kernels: one wave per SIMD with all 512 registers; the accumulation half holds the resident layer-1 output
         # (MFMA B operands), so the accumulators go to the architectural half ("VGPR form")
         # (compiled in three parts -- see the unit's PSTL_C2_PART: its twelve kernel instantiations take minutes in one piece)
         ("chain2_kernels.hip", ["-Xarch_device", "-mllvm=-amdgpu-mfma-vgpr-form", "-DPSTL_C2_PART=0"], "chain2_kernels.o"),
         ("chain2_kernels.hip", ["-Xarch_device", "-mllvm=-amdgpu-mfma-vgpr-form", "-DPSTL_C2_PART=1"], "chain2_kernels_p1.o"),
         ("chain2_kernels.hip", ["-Xarch_device", "-mllvm=-amdgpu-mfma-vgpr-form", "-DPSTL_C2_PART=2"], "chain2_kernels_p2.o"),
         ("diversity_kernels.hip", ["-ffp-contract=off"]), ("stl_program.hip", ["-ffp-contract=off"]),
         # adam_kernels: torch.optim.Adam's float32 update, every operation rounded on its own (csrc/adam_core.hpp)
         ("adam_kernels.hip", ["-ffp-contract=off"])]
LINK_LIBS = []   # no vendor BLAS: every kernel of the library is in csrc/


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(HERE, "..", "include", "pstl_hip.h"))
    objs, running = [], []
    for unit in UNITS:      # the units compile side by side
        src, extra = unit[0], unit[1]
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, unit[2] if len(unit) > 2 else src.replace(".hip", ".o"))
        if force or _newer(o, [s] + headers):
            cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC"] + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            # (each compiler in a process group of its own: hipcc is a wrapper around clang children, and killing the wrapper alone
            # would leave them writing an object file behind a failed build)
            running.append((cmd, subprocess.Popen(cmd, start_new_session=True), o, time.time()))
        objs.append(o)
    times, pending = {}, list(running)
    while pending:          # (polled, so that the time printed for a unit is its own and not that of the slowest one before it)
        time.sleep(0.2)
        for item in list(pending):
            cmd, proc, o, t0 = item
            rc = proc.poll()
            if rc is None:
                continue
            pending.remove(item)
            if rc != 0:
                for _, other, _, _ in running:      # (only the process groups started here; never by pattern)
                    if other.poll() is None:
                        try:
                            os.killpg(other.pid, signal.SIGKILL)
                        except OSError:
                            pass
                        other.wait()
                for _, _, oo, _ in running:          # no output of the failed batch may pass for up to date next time
                    if os.path.exists(oo):
                        os.remove(oo)
                raise subprocess.CalledProcessError(rc, cmd)
            times[os.path.basename(o)] = time.time() - t0
    if verbose and times:
        print("compiled: " + ", ".join("%s %.0f s" % kv for kv in sorted(times.items())), flush=True)
    if force or _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + LINK_LIBS + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
