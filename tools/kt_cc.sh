#!/bin/bash
# kernel trace of the last CC enact: bash tools/kt_cc.sh <scale>
root=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ktcc
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktcc -o kt -- python3 $root/tools/one_cc.py $1 2 > $root/gpurun_out/kt_cc.log 2>&1
python3 $root/tools/kt_print.py $(find /tmp/ktcc -name "*kernel_trace.csv" | head -1) HookInit > $root/gpurun_out/kt_cc.txt
cat $root/gpurun_out/kt_cc.txt | cut -c1-120
