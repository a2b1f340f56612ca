#!/bin/bash
# bench each variant library: bash tools/variant_sweep.sh name1 name2 ...  (plus "base" for the in-tree build)
for v in "$@"; do
  if [ "$v" = base ]; then unset GUNROCK_LIB_PATH; else export GUNROCK_LIB_PATH=$(pwd)/tools/variants/$v.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --skip-topdown-leg > gpurun_out/v_$v.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/v_$v.log; continue; }
  tail -1 gpurun_out/v_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['enact_ms_per_step'], d['roofline']['by_kernel_ms'])"
done
