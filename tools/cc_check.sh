#!/bin/bash
python -m pytest tests/test_cc_gpu.py tests/test_examples_gpu.py -q -x 2>&1 | tail -1
for i in 1 2; do python bench.py --primitive cc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['enact_ms_per_step'], d['roofline']['frac'], d['parity_vs_oracle'])"; done
python tools/fuzz_others.py 30 1010 2>&1 | tail -1
