#!/bin/bash
# A/B of library builds on one box: tools/ab_libs.sh <family,family,...> lib1.so lib2.so ...  ("-" = the in-tree library)
# prints ms per step and the per-step time of the named kernel families for each build.
fams=$1; shift
for l in "$@"; do
  if [ "$l" = "-" ]; then unset MISAMD_LIBRARY; else export MISAMD_LIBRARY=$PWD/$l; fi
  timeout -k 10 150 python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null > /tmp/ab_line.json || exit 1
  python3 - "$l" "$fams" <<'P'
import json, sys
d = json.loads(open('/tmp/ab_line.json').read().strip().splitlines()[-1])
k = d['kernels']
print(sys.argv[1], d['ms_per_step'], ' '.join('%s=%.4f' % (f, k[f]['ms_per_step']) for f in sys.argv[2].split(',') if f in k), flush=True)
P
done
