#!/bin/bash
# CC check: tests, bench, fuzz, kernel timeline of one scale-24 run: bash tools/cc_rounds.sh
python -m pytest tests/test_cc_gpu.py tests/test_examples_gpu.py -q -x 2>&1 | tail -1
for r in 1 1; do python bench.py --primitive cc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['enact_ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_own_sweeps'), d['parity_vs_oracle'])"; done
python tools/fuzz_others.py 40 2020 2>&1 | tail -1
bash tools/kt_cc.sh 24
