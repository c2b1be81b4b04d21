#!/usr/bin/env python3
"""HBM traffic per kernel family from the two PMC passes of tools/pmc_bench.sh.

traffic = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KiB): on gfx950 FETCH_SIZE counts the
128-byte requests of wide streaming reads at 64 bytes, WRITE_SIZE is exact for 16-byte-per-lane
stores.  usage: pmc_traffic.py pass1.db pass2.db > profiles/<name>.json"""
import json
import re
import sqlite3
import sys
from collections import defaultdict

FAMILIES = [
    ("conv3x3_fwd", r"conv3x3_mfma_kernel|conv3x3_first_kernel|conv3x3_direct_kernel"),
    ("conv3x3_wgrad", r"conv3x3_wgrad_mfma_kernel|conv3x3_wgrad_direct_kernel|wgrad_reduce_kernel"),
    ("bn_relu_bwd_dz", r"bn_relu_bwd_dz_kernel"),
    ("bn_relu_bwd_reduce", r"BnBwdReduce"),
    ("maxpool2x2_bwd", r"maxpool_bwd_kernel"),
    ("maxpool2x2_fwd", r"maxpool_fwd_kernel"),
    ("upsample2x_fwd", r"up_fwd_tiled_kernel"),
    ("upsample2x_bwd", r"up_bwd_tiled_kernel"),
    ("conv1x1_fwd", r"conv1x1_fwd_kernel|conv1x1_fwd_cv_kernel"),
    ("conv1x1_bwd", r"conv1x1_bwd_kernel"),
    ("loss_metrics_fwd", r"loss_metrics_kernel"),
    ("loss_bwd", r"loss_bwd_kernel"),
]


def main():
    per = defaultdict(lambda: defaultdict(float))
    count = defaultdict(int)
    for path in sys.argv[1:]:
        cur = sqlite3.connect(path).cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        for k, c, v, n in cur.execute(f"select {kcol}, counter_name, sum(value), count(distinct dispatch_id) "
                                      f"from counters_collection group by {kcol}, counter_name"):
            for fam, pat in FAMILIES:
                if re.search(pat, k):
                    per[fam][c] += v
                    if c == "FETCH_SIZE":
                        count[fam] += n
                    break
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE over `bench.py --steps 2 --warmup 1` (3 steps)",
           "formula": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024", "families": {}}
    for fam, _ in FAMILIES:
        if count[fam]:
            d = per[fam]
            tot = (2 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024
            out["families"][fam] = {"kernel_launches": count[fam],
                                    "fetch_kib": round(d.get("FETCH_SIZE", 0.0), 1),
                                    "write_kib": round(d.get("WRITE_SIZE", 0.0), 1),
                                    "traffic_bytes_per_step": round(tot / 3),
                                    "traffic_bytes_per_kernel_launch": round(tot / count[fam])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
