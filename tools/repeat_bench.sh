#!/bin/bash
# Runs the bench several times per setting of the second weight-gradient stream and prints
# value / final loss / mean dice of every run (a run-to-run difference in the last two = a race).
#   tools/repeat_bench.sh "1 1 1 0 0 0" [bench flags]
set -u
modes=${1:-"1 1 1 0 0 0"}; shift || true
for v in $modes; do
  MIS_WGRAD_STREAM=$v timeout -k 10 150 python bench.py --no-cpu-baseline "$@" 2>/dev/null > /tmp/rb.json || { echo "side_stream $v: run failed (non-finite loss or error)"; continue; }
  python - "$v" <<'EOF'
import json, sys
d = json.load(open("/tmp/rb.json"))
print("side_stream", sys.argv[1], "value", d["value"], "loss", d["final_loss"], "dice", d["mean_dice"], flush=True)
EOF
done
