#!/usr/bin/env python3
"""bench.py with attributes of medical_image_segmentation_amd.functional overridden first:
    python tools/bench_with.py BN_SUMS_MIN_NUMEL=30000000 -- --steps 20 --no-cpu-baseline"""
import ast
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medical_image_segmentation_amd import functional as F_  # noqa: E402

args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
for kv in args[:split]:
    k, v = kv.split("=", 1)
    assert hasattr(F_, k), k
    setattr(F_, k, ast.literal_eval(v))
sys.argv = ["bench.py"] + args[split + 1:]
import bench  # noqa: E402

bench.main()
