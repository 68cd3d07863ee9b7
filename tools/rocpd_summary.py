"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite) outputs into small text files for profiles/.

Usage: python tools/rocpd_summary.py <results.db> [--counters]
  default    : per-kernel count / total / avg / min / max duration (the `--stats` view)
  --counters : per-kernel mean of every collected PMC counter (summed over dimensions per dispatch)
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    if "--counters" in sys.argv:
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        rows = cur.execute(
            "select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection "
            "group by kernel_name, counter_name, dispatch_id").fetchall()
        agg = {}
        for k, c, d, v in rows:
            agg.setdefault((k, c), []).append(v)
        print("%-28s %-24s %10s %18s" % ("kernel", "counter", "dispatches", "mean_per_dispatch"))
        for (k, c), vals in sorted(agg.items()):
            print("%-28s %-24s %10d %18.6g" % (k[:28], c, len(vals), sum(vals) / len(vals)))
        return
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-40s %6s %14s %14s %14s %14s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for name, n, tot, avg, mn, mx in rows:
        print("%-40s %6d %14d %14.0f %14d %14d %6.2f%%" % (name[:40], n, tot, avg, mn, mx, 100.0 * tot / total))


if __name__ == "__main__":
    main()
