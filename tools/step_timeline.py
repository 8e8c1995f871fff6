#!/usr/bin/env python3
"""Ordered kernel list of ONE step of a rocprofv3 kernel trace (the last step = the dispatches after the last
`pair_l2`-free gap): python tools/step_timeline.py TRACE.csv [marker-kernel-substring]
Prints per dispatch: start offset (us), duration (us), gap to the previous dispatch's end (us), short name."""
import csv
import sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from prof_summary import short

csv.field_size_limit(1 << 30)
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Kind", "KERNEL_DISPATCH") != "KERNEL_DISPATCH":
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
marker = sys.argv[2] if len(sys.argv) > 2 else "embed"
# a step starts at the first dispatch whose name contains the marker after a non-marker dispatch
starts = [i for i, r in enumerate(rows) if marker in r[2] and (i == 0 or marker not in rows[i - 1][2])]
if len(starts) < 3:
    print("marker %r found %d times" % (marker, len(starts)))
    sys.exit(1)
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(starts) - 2      # which step (index into the marker occurrences)
print("# %d marker occurrences; step %d" % (len(starts), k))
a, b = starts[k], starts[k + 1]
t0 = rows[a][0]
prev_end = t0
ktime = 0
for s, e, n in rows[a:b]:
    print("%9.1f %8.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n))
    prev_end = e
    ktime += e - s
print("# %d dispatches, span %.1f us, kernel time %.1f us" % (b - a, (rows[b][0] - t0) / 1e3, ktime / 1e3))
