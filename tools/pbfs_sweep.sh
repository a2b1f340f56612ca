#!/bin/bash
# forced one-rank partitioned bench with different level-loop options: bash tools/pbfs_sweep.sh "lite_factor=0" "lite_factor=230,alpha=30" ...
for o in "$@"; do
  GUNROCK_FORCE_PARTITIONED=1 GUNROCK_PBFS_OPTIONS="$o" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/pb.log 2>&1 || { echo "[$o] FAILED"; tail -3 gpurun_out/pb.log; continue; }
  tail -1 gpurun_out/pb.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$o]', d['ms_per_step'], d['value'], d['config'].get('count_only_levels_per_search'))"
done
