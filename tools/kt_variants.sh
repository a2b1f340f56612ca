#!/bin/bash
# kernel trace of one BFS per variant library: bash tools/kt_variants.sh <scale> <mode> <src idx> v1 v2 ...
scale=$1; mode=$2; k=$3; shift 3
for v in "$@"; do
  if [ "$v" = base ]; then unset GUNROCK_LIB_PATH; else export GUNROCK_LIB_PATH=$(pwd)/tools/variants/$v.so; fi
  bash tools/kt_one.sh $scale $mode $k v_$v
  echo "== $v"; grep -v "rocclr\|Publish" gpurun_out/kt_v_$v.txt
done
