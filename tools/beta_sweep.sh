#!/bin/bash
# BFS bench line per beta (bottom-up -> top-down switch: frontier vertices * beta < n): bash tools/beta_sweep.sh <scale> b1 b2 ...
scale=$1; shift
for b in "$@"; do
  python bench.py --scale $scale --beta $b --no-cpu-baseline --skip-topdown-leg 2>/dev/null > /tmp/beta.json
  python -c "import json;d=json.load(open('/tmp/beta.json'));print('scale', $scale, 'beta', $b, 'GTEPS', round(d['value']/1e3,1), 'ms/step', d['ms_per_step'], 'enact ms', d['enact_ms_per_step'])"
done
