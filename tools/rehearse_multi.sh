#!/bin/bash
# the N-rank bench path end to end on ONE GPU (ranks share it, gloo host-staged exchanges): bash tools/rehearse_multi.sh <ranks> [scale]
n=${1:-2}; scale=${2:-20}
GUNROCK_DIST_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
  --master-port $((29560 + n)) bench.py --gpus $n --scale $scale --steps 10 --warmup 2 > gpurun_out/mg$n.log 2>&1
echo "rc=$?"
tail -1 gpurun_out/mg$n.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], 'ranks:', d['value'], 'MTEPS', d['ms_per_step'], 'ms/step parity', d['parity_vs_oracle'], d['config']['transport'], 'count-only levels/search', d['config']['count_only_levels_per_search'], 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])" || tail -20 gpurun_out/mg$n.log
