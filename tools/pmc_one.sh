#!/bin/bash
# PMC passes over one BFS: bash tools/pmc_one.sh <scale> <mode> <source index|-1> <tag> "<counters pass 1>" ["<counters pass 2>" ...]
# -> gpurun_out/pmc_<tag>.txt : per dispatch of the last BFS, counter values (kernel-trace is NOT combined with --pmc)
root=$(pwd); scale=$1; mode=$2; k=$3; tag=$4; shift 4
cd /tmp && export TMPDIR=/tmp
: > $root/gpurun_out/pmc_$tag.txt
i=0
for pass in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${tag}_$i
  echo "pass $i: $pass"
  rocprofv3 --pmc $pass --kernel-include-regex "${PMC_KERNELS:-.*}" --output-format csv -d /tmp/pmc_${tag}_$i -o p -- python3 $root/tools/one_bfs.py $scale $mode $k 1 > $root/gpurun_out/pmc_${tag}_$i.log 2>&1
  python3 $root/tools/pmc_last_bfs.py $(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1) >> $root/gpurun_out/pmc_$tag.txt
  echo "pass $i done"
done
