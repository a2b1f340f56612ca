#!/bin/bash
# Round profile: kernel-trace stats and two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the default bench command.
# Usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
set -e
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o st -- python3 $root/bench.py --no-cpu-baseline > $out/bench_under_rocprof.log 2>&1
rm -rf /tmp/p_stats_old; cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
grep -h "^{" $out/bench_under_rocprof.log | tail -1 > $out/bench_line_under_rocprof.json || true
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -o f -- python3 $root/bench.py --steps 10 --warmup 1 --no-cpu-baseline --skip-topdown-leg > $out/pmc_fetch.log 2>&1
python3 $root/tools/pmc_summary.py $(find /tmp/p_fetch -name "*counter_collection.csv" | head -1) FETCH_SIZE > $out/pmc_fetch_size.json
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -o w -- python3 $root/bench.py --steps 10 --warmup 1 --no-cpu-baseline --skip-topdown-leg > $out/pmc_write.log 2>&1
python3 $root/tools/pmc_summary.py $(find /tmp/p_write -name "*counter_collection.csv" | head -1) WRITE_SIZE > $out/pmc_write_size.json
echo "write done"
python3 -c "import sys,json; sys.path.insert(0,'$root'); import bench; json.dump({'source_sha': bench.source_fingerprint(), 'command': 'python3 bench.py [--no-cpu-baseline] (stats) / --steps 10 --warmup 1 --no-cpu-baseline --skip-topdown-leg (pmc)'}, open('$out/profile_meta.json','w'))"
echo "meta done"
