#!/bin/bash
# SSSP bench line per delta factor: bash tools/sssp_delta_sweep.sh <scale> f1 f2 ...
scale=$1; shift
for df in "$@"; do
  python bench.py --primitive sssp --scale $scale --delta-factor $df --no-cpu-baseline 2>/dev/null > /tmp/sssp_df.json
  python -c "import json;d=json.load(open('/tmp/sssp_df.json'));print('delta_factor', $df, 'MTEPS', d['value'], 'enact ms', d['enact_ms_per_step'])"
done
