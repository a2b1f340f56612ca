#!/bin/bash
# run tools/level_trace.py <scale> <mode> for each variant library: bash tools/exp_trace.sh <scale> <mode> v1 v2 ...
scale=$1; mode=$2; shift 2
for v in "$@"; do
  if [ "$v" = base ]; then unset GUNROCK_LIB_PATH; else export GUNROCK_LIB_PATH=$(pwd)/tools/variants/$v.so; fi
  echo "== $v"
  timeout -k 10 120 python tools/level_trace.py $scale $mode 2>&1 | grep -v amdgpu.ids
done
