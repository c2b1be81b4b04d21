#!/usr/bin/env python3
"""Per-kernel mean PMC counter values from rocprofv3 rocpd sqlite files.
usage: rocpd_pmc.py pass1.db [pass2.db ...]"""
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    out = defaultdict(dict)
    for path in sys.argv[1:]:
        con = sqlite3.connect(path)
        cur = con.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        view = "counters_collection" if "counters_collection" in tabs else None
        if view is None:
            print(path, "no counters view; tables:", [t for t in tabs if "pmc" in t or "counter" in t][:10])
            continue
        cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        rows = cur.execute(f"select {kcol}, counter_name, sum(value), count(distinct dispatch_id) from {view} "
                           f"group by {kcol}, counter_name").fetchall()
        for k, c, v, n in rows:
            short = re.sub(r"^void ", "", k).replace("(anonymous namespace)::", "").split("(")[0][:90]
            out[short][c] = v / max(n, 1)
            out[short]["_n"] = n
    for k, d in out.items():
        print(k, "dispatches", d.pop("_n"))
        for c, v in sorted(d.items()):
            print(f"    {c:28s} {v:16.1f}")


if __name__ == "__main__":
    main()
