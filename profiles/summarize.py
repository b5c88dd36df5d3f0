"""Turns a rocprofv3 `--kernel-trace --stats` result (rocpd sqlite .db, or *_kernel_stats.csv)
into the small text summary committed under profiles/.
    python profiles/summarize.py <results.db|kernel_stats.csv> <out.txt> "<command line that was profiled>" [build fingerprint]
The fingerprint (bench.build_fingerprint() of the profiled build) goes into the header: bench.py quotes a kernel's average from
this file only when it matches the loaded build's.
"""
import csv
import sqlite3
import sys


def rows_from_db(path):
    cur = sqlite3.connect(path).cursor()
    return [(n, int(c), float(t), float(a), float(p)) for n, c, t, a, p in
            cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels")]


def rows_from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return out


def main():
    src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
    fp = sys.argv[4] if len(sys.argv) > 4 else ""
    rows = rows_from_db(src) if src.endswith(".db") else rows_from_csv(src)
    with open(dst, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n# command: %s\n" % cmd)
        if fp:
            f.write("# build_fingerprint: %s\n" % fp)
        f.write("%-90s %7s %12s %10s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "%"))
        for n, c, t, a, p in rows:
            f.write("%-90s %7d %12.1f %10.2f %7.2f\n" % (n[:90], c, t, a, p))


if __name__ == "__main__":
    main()
