#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: per-kernel totals over the last `n` occurrences of the anchor kernel (one per planning
step) and the idle time between consecutive kernels.  python tools/dbg/trace_gaps.py trace.csv [n_steps]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor = [i for i, r in enumerate(rows) if "k_prepare" in r["Kernel_Name"]]
lo = anchor[-n - 1]
hi = anchor[-1]
seg = rows[lo:hi]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(seg, seg[1:])]
print("steps %d: kernels/step %.1f, busy %.1f us/step, span %.1f us/step, gaps: total %.1f us/step, median %.2f us, >20us: %d"
      % (n, len(seg) / n, busy / 1e3 / n, span / 1e3 / n, sum(gaps) / 1e3 / n, sorted(gaps)[len(gaps) // 2] / 1e3,
         sum(1 for g in gaps if g > 20000)))
per = collections.defaultdict(lambda: [0, 0])
for r in seg:
    m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
    nm = m.group(1) if m else r["Kernel_Name"][:50]
    per[nm][0] += 1
    per[nm][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for nm, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("  %-62s %5.1f calls/step %8.2f us each %8.1f us/step" % (nm[:62], c / n, t / 1e3 / c, t / 1e3 / n))
