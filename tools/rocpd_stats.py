"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace run
whose output is the SQLite database (`*_results.db`) rather than CSV files.

    python tools/rocpd_stats.py RESULTS.db OUT.csv
"""
import csv
import re
import sqlite3
import sys


def kernel_stats(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, count(*), sum(end - start), avg(end - start), min(end - start), "
        "max(end - start) from kernels group by name order by 3 desc").fetchall()
    total = float(sum(r[2] for r in rows)) or 1.0
    out = []
    for name, calls, tot, avg, lo, hi in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        out.append((name, calls, tot, round(avg, 1), round(100.0 * tot / total, 2), lo, hi))
    return out


def main():
    rows = kernel_stats(sys.argv[1])
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        w.writerows(rows)
    for name, calls, tot, avg, pct, lo, hi in rows[:14]:
        print("  %-72s %6d %10.2f us %6.2f %%" % (name[:72], calls, avg / 1e3, pct))


if __name__ == "__main__":
    main()
