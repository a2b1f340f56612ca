#!/bin/bash
# kernel trace of an arbitrary python tool: bash tools/kt_cmd.sh <tag> <script> args...  -> gpurun_out/kt_<tag>.txt (last search)
root=$(pwd); tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -o kt -- python3 $root/"$@" > $root/gpurun_out/kt_$tag.log 2>&1
python3 $root/tools/kt_print.py $(find /tmp/kt_$tag -name "*kernel_trace.csv" | head -1) ${KT_ANCHOR:-ResetKernel} > $root/gpurun_out/kt_$tag.txt
