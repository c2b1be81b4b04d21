#!/bin/bash
set -e
for d in "" "--dist"; do for s in 1 0; do MIS_WGRAD_STREAM=$s timeout -k 10 100 python tools/host_step_time.py $d 2>/dev/null | tail -1; done; done
