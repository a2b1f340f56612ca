#!/bin/bash
python -m pytest tests/test_bc_gpu.py -q -x 2>&1 | tail -1
for i in 1 2; do python bench.py --primitive bc --scale 22 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['run_ms_per_step'], d['value'], d['roofline']['frac'], d['parity_vs_oracle'])"; done
python tools/fuzz_others.py 30 909 2>&1 | tail -1
