#!/usr/bin/env python3
"""Static instruction counts per kernel of one translation unit (device assembly via hipcc -S): total, VALU, transcendental,
s_nop, VGPRs, scratch.  python tools/isa_counts.py stl_kernels.hip [filter]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pstl_diffusion_policy_amd.build import UNITS, CSRC  # noqa: E402


def main():
    unit = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    extra = next(u[1] for u in UNITS if u[0] == unit)
    out = "/tmp/isa_%s.s" % unit.replace(".hip", "")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17"] + extra + sys.argv[3:] +
                          ["--cuda-device-only", "-S", os.path.join(CSRC, unit), "-o", out], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for (i, name), (j, _) in zip(starts, starts[1:] + [(len(lines), "")]):
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = dn.replace("pstl::(anonymous namespace)::", "").replace("void ", "")
        if flt not in dn:
            continue
        body = lines[i:j]
        ins = [l.strip().split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = collections.Counter(ins)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        trans = sum(v for k, v in c.items() if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", k))
        # the compiler's resource summary follows the kernel descriptor ("; NumVgprs: 21" ...)
        txt = "\n".join(body)
        vg = re.search(r"; NumVgprs: (\d+)", txt)
        ag = re.search(r"; NumAgprs: (\d+)", txt)
        sc = re.search(r"; ScratchSize: (\d+)", txt)
        oc = re.search(r"; Occupancy: (\d+)", txt)
        print("%6d valu %6d trans %4d nop %4d vgpr %4s agpr %3s scratch %4s occ %s  %s" % (
            len(ins), valu, trans, c.get("s_nop", 0), vg.group(1) if vg else "?", ag.group(1) if ag else "?",
            sc.group(1) if sc else "?", oc.group(1) if oc else "?", dn[:100]))


if __name__ == "__main__":
    main()
