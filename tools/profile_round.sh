#!/bin/bash
# Round profile: kernel-trace stats and separate PMC passes (FETCH_SIZE; WRITE_SIZE; TCC_HIT_sum + TCC_MISS_sum) of the default bench
# command (headline leg only: --no-secondary), plus a TCC pass of the top-down-only mode.
# Usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
set -e
tag=${1:-r03}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o st -- python3 $root/bench.py --no-cpu-baseline --no-secondary > $out/bench_under_rocprof.log 2>&1
rm -rf /tmp/p_stats_old; cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
grep -h "^{" $out/bench_under_rocprof.log | tail -1 > $out/bench_line_under_rocprof.json || true
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -o f -- python3 $root/bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-secondary --skip-topdown-leg > $out/pmc_fetch.log 2>&1
python3 $root/tools/pmc_summary.py $(find /tmp/p_fetch -name "*counter_collection.csv" | head -1) FETCH_SIZE > $out/pmc_fetch_size.json
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -o w -- python3 $root/bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-secondary --skip-topdown-leg > $out/pmc_write.log 2>&1
python3 $root/tools/pmc_summary.py $(find /tmp/p_write -name "*counter_collection.csv" | head -1) WRITE_SIZE > $out/pmc_write_size.json
echo "write done"
# L2 (TCC) hit / miss per kernel: the evidence row of SURVEY 8(d) behind "bound by the L2 random-probe rate, not by HBM"
rm -rf /tmp/p_tcc /tmp/p_tcc_td
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/p_tcc -o t -- python3 $root/bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-secondary --skip-topdown-leg > $out/pmc_tcc.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/p_tcc_td -o t -- python3 $root/bench.py --steps 4 --warmup 1 --traversal-mode 0 --no-cpu-baseline --no-secondary > $out/pmc_tcc_td.log 2>&1
python3 $root/tools/pmc_tcc.py $(find /tmp/p_tcc -name "*counter_collection.csv" | head -1) $(find /tmp/p_tcc_td -name "*counter_collection.csv" | head -1) > $out/pmc_tcc.json
echo "tcc done"
python3 -c "import sys,json; sys.path.insert(0,'$root'); import bench; json.dump({'source_sha': bench.source_fingerprint(), 'command': 'python3 bench.py [--no-cpu-baseline] (stats) / --steps 10 --warmup 1 --no-cpu-baseline --no-secondary --skip-topdown-leg (pmc)'}, open('$out/profile_meta.json','w'))"
echo "meta done"
