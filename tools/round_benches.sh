#!/bin/bash
# the secondary bench lines kept under profiles/: bash tools/round_benches.sh <tag>   (on the GPU box, from the repo root)
tag=${1:-r03}; out=gpurun_out/benches_$tag; mkdir -p $out
run() { name=$1; shift; timeout -k 10 600 python bench.py "$@" 2> $out/$name.err | tail -1 > $out/${tag}_bench_$name.json || echo "$name FAILED"; python -c "
import json,sys; d=json.load(open('$out/${tag}_bench_$name.json')); print('$name', d['value'], d.get('ms_per_step'), d.get('enact_ms_per_step'), d['roofline']['frac'], d.get('parity_vs_oracle'))"; }
run scale22 --scale 22 --no-secondary
run scale26 --scale 26 --steps 20 --warmup 2 --cpu-baseline-runs 1 --no-secondary
run cc_scale24 --primitive cc
run sssp_lj --primitive sssp
run sssp_scale22 --primitive sssp --graph rmat --scale 22
run bc_scale22 --primitive bc --scale 22
run pr_scale22 --primitive pr --scale 22
run pr_scale24 --primitive pr --scale 24
GUNROCK_FORCE_PARTITIONED=1 timeout -k 10 600 python bench.py --cpu-baseline-runs 1 2> $out/fp.err | tail -1 > $out/${tag}_bench_forced_partition_1rank.json; python -c "
import json; d=json.load(open('$out/${tag}_bench_forced_partition_1rank.json')); print('forced_partition', d['value'], d['ms_per_step'], d['parity_vs_oracle'], d['config']['transport'], d['cpu_baseline'])"
