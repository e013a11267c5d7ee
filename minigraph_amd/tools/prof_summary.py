#!/usr/bin/env python
"""Turn a rocprofv3 `--kernel-trace --stats` result (sqlite .db or *_kernel_stats.csv) into the short
text summary committed under profiles/.  usage: prof_summary.py <results.db|kernel_stats.csv> [title]"""
import csv
import sqlite3
import sys


def rows(path):
    if path.endswith(".db"):
        cur = sqlite3.connect(path).cursor()
        for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            yield name, int(calls), float(tot), float(avg), float(pct)
    else:
        for r in csv.DictReader(open(path)):
            yield r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])


def main():
    path, title = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    print("# rocprofv3 --kernel-trace --stats  %s" % title)
    print("# %-40s %8s %14s %14s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows(path):
        print("%-42s %8d %14.1f %14.1f %7.2f" % (name.split("(")[0][:42], calls, tot, avg, pct))


if __name__ == "__main__":
    main()
