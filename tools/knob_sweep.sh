#!/bin/bash
# bench the in-tree library with different bench.py knobs: bash tools/knob_sweep.sh "<args1>" "<args2>" ...
i=0
for a in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py --no-cpu-baseline --skip-topdown-leg --no-secondary $a > gpurun_out/k_$i.log 2>&1 || { echo "[$a] FAILED"; tail -3 gpurun_out/k_$i.log; continue; }
  tail -1 gpurun_out/k_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$a]', d['value'], d['ms_per_step'], d['enact_ms_per_step'], d['roofline']['frac'], d['roofline']['by_kernel_ms'])"
done
