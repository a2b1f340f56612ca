#!/bin/bash
# rocprofv3 kernel stats of any bench.py invocation: bash tools/kstats.sh <tag> <bench args...>  -> gpurun_out/kstats_<tag>.csv
tag=$1; shift
root=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o st -- python3 $root/bench.py "$@" > $root/gpurun_out/kstats_$tag.log 2>&1
cp $(find /tmp/ks_$tag -name "*kernel_stats.csv" | head -1) $root/gpurun_out/kstats_$tag.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$root/gpurun_out/kstats_$tag.csv")))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:14]:
    print(r['Name'].split('(')[0][-90:], r['Calls'], round(float(r['TotalDurationNs'])/int(r['Calls'])/1e3,1), 'us avg', round(float(r['TotalDurationNs'])/1e6,2), 'ms total')
PY
