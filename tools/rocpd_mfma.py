#!/usr/bin/env python3
"""Matrix-core utilisation per kernel from one rocprofv3 --pmc pass that collected
SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_BF16 and GRBM_GUI_ACTIVE (rocpd sqlite file).

  busy %   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
             (the busy counter adds up SIMD cycles: 16 per v_mfma_f32_16x16x32_bf16, checked against
             SQ_INSTS_MFMA; GRBM_GUI_ACTIVE is the sum of the 8 XCDs' active cycles)
  TFLOP/s  = MOPS_BF16 * 512 / (GRBM_GUI_ACTIVE / 8 / clock) at the clock given (default 2.4 GHz:
             an upper bound, the chip clocks lower under a profiler)

usage: rocpd_mfma.py results.db [clock_ghz]   -> CSV on stdout"""
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    con = sqlite3.connect(sys.argv[1])
    clock = float(sys.argv[2]) if len(sys.argv) > 2 else 2.4
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    acc = defaultdict(lambda: defaultdict(float))
    nd = defaultdict(set)
    for k, c, v, d in cur.execute(f"select {kcol}, counter_name, value, dispatch_id from counters_collection"):
        short = re.sub(r"^void ", "", k).replace("(anonymous namespace)::", "").split("(")[0][:100]
        acc[short][c] += v
        nd[short].add(d)
    print("kernel,dispatches,mfma_busy_pct,bf16_tflops_at_%.1fGHz,active_us_at_%.1fGHz_per_dispatch" % (clock, clock))
    rows = []
    for k, d in acc.items():
        busy, act = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("GRBM_GUI_ACTIVE", 0.0)
        if busy <= 0 or act <= 0:
            continue
        cyc = act / 8.0
        t = cyc / (clock * 1e9)
        rows.append((busy, k, len(nd[k]), 100.0 * busy / (cyc * 1024.0),
                     d.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512 / t / 1e12, 1e6 * t / len(nd[k])))
    for _, k, n, pct, tf, us in sorted(rows, reverse=True):
        print(f"{k},{n},{pct:.1f},{tf:.0f},{us:.1f}")


if __name__ == "__main__":
    main()
