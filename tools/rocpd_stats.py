#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd SQLite database (the default output of ROCm 7.2's rocprofv3
--kernel-trace), equivalent to the `--stats` kernel summary:  python tools/rocpd_stats.py results.db [last_n_steps]"""
import re
import sqlite3
import sys


def main(path, out=sys.stdout):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    ksym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in ksym_cols else "kernel_name"
    q = ("select s.%s, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
         "on d.kernel_id = s.id" % name_col)
    rows = cur.execute(q).fetchall()
    agg = {}
    for name, st, en in rows:
        name = re.sub(r"\(.*", "", name)
        name = re.sub(r"^void ", "", name)
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        dur = en - st
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    print("# kernel statistics from %s (%d dispatches, %.3f ms total kernel time; columns of rocprofv3 --stats)" %
          (path, len(rows), total / 1e6), file=out)
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs", file=out)
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (name[:110], a[0], a[1], a[1] / a[0], 100.0 * a[1] / total, a[2], a[3]),
              file=out)


if __name__ == "__main__":
    main(sys.argv[1])
