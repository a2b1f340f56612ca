#!/bin/bash
# kernel trace of one BFS: bash tools/kt_one.sh <scale> <mode> <source index|-1> <tag>   -> gpurun_out/kt_<tag>.txt
root=$(pwd); scale=$1; mode=$2; k=$3; tag=$4
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -o kt -- python3 $root/tools/one_bfs.py $scale $mode $k 2 > $root/gpurun_out/kt_$tag.log 2>&1
python3 $root/tools/kt_print.py $(find /tmp/kt_$tag -name "*kernel_trace.csv" | head -1) > $root/gpurun_out/kt_$tag.txt
