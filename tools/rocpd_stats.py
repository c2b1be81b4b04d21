#!/usr/bin/env python3
"""Per-kernel statistics (calls, total/avg duration, share) from a rocprofv3 rocpd sqlite file.
usage: rocpd_stats.py results.db [steps] > stats.csv"""
import re
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end - start), min(end - start), max(end - start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    hdr = "kernel,calls,total_us,avg_us,min_us,max_us,percent"
    if steps:
        hdr += ",us_per_step,calls_per_step"
    print(hdr)
    for name, n, tot, mn, mx in rows:
        short = re.sub(r"^void ", "", name)
        short = re.sub(r"\(anonymous namespace\)::", "", short)
        short = short.split("(")[0] if not short.startswith("(") else short
        line = f"\"{short[:140]}\",{n},{tot / 1e3:.1f},{tot / n / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100 * tot / total:.2f}"
        if steps:
            line += f",{tot / 1e3 / steps:.1f},{n / steps:.1f}"
        print(line)
    print(f"\"TOTAL\",{sum(r[1] for r in rows)},{total / 1e3:.1f},,,,100.0" + (f",{total / 1e3 / steps:.1f}," if steps else ""))


if __name__ == "__main__":
    main()
