#!/bin/bash
# One PMC pass per counter list over an arbitrary python tool (never combined with a trace):
#   bash tools/pmc_cmd.sh <tag> "<counters pass 1>[;<counters pass 2>...]" <script> args...  -> gpurun_out/pmc_<tag>.txt
root=$(pwd); tag=$1; passes=$2; shift 2
cd /tmp && export TMPDIR=/tmp
: > $root/gpurun_out/pmc_$tag.txt
i=0
IFS=';' read -ra LIST <<< "$passes"
for pass in "${LIST[@]}"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${tag}_$i
  echo "pass $i: $pass"
  timeout -k 10 ${PMC_TIMEOUT:-150} rocprofv3 --pmc $pass --kernel-include-regex "${PMC_KERNELS:-.*}" --output-format csv -d /tmp/pmc_${tag}_$i -o p -- python3 $root/"$@" > $root/gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pass $i rc $?"
  f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $root/tools/pmc_rows.py $f >> $root/gpurun_out/pmc_$tag.txt
done
