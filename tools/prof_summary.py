#!/usr/bin/env python3
"""Summaries of rocprofv3 CSV output directories (kernel trace -> the columns of `--stats`; counter collection -> mean
counter value per dispatch of every kernel):   python tools/prof_summary.py DIR [--min-pct 0.3]"""
import collections
import csv
import glob
import os
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:                      # cut at the first '(' outside template brackets (= the argument list)
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()[:110]


def trace_stats(path, out):
    agg = collections.OrderedDict()
    n = 0
    for row in csv.DictReader(open(path)):
        if row.get("Kind", "KERNEL_DISPATCH") != "KERNEL_DISPATCH":
            continue
        k = short(row["Kernel_Name"])
        d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        a = agg.setdefault(k, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        n += 1
    tot = sum(a[1] for a in agg.values()) or 1
    print("# %s: %d dispatches, %.3f ms total kernel time" % (path, n, tot / 1e6), file=out)
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs", file=out)
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (k, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, a[2], a[3]), file=out)


def counter_stats(path, out):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        k = (short(row["Kernel_Name"]), row["Counter_Name"])
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    print("# %s: mean counter value per dispatch" % path, file=out)
    print("Name,Counter,Dispatches,MeanPerDispatch,Total", file=out)
    for (k, c), (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print('"%s",%s,%d,%.6g,%.6g' % (k, c, n, v / n, v), file=out)


if __name__ == "__main__":
    root = sys.argv[1]
    for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        if "pmc_" in f:
            continue        # timing under counter collection is not representative; the counter table follows
        trace_stats(f, sys.stdout)
        print()
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        counter_stats(f, sys.stdout)
        print()
