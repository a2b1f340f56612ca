#!/bin/bash
# SSSP parity tests + the config-3 bench line twice: bash tools/sssp_check.sh
python -m pytest tests/test_sssp_gpu.py -q -x 2>&1 | tail -1
for i in 1 2; do python bench.py --primitive sssp --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['enact_ms_per_step'], d['value'], d['roofline']['frac'])"; done
python bench.py --primitive sssp --no-cpu-baseline --delta-factor 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('delta_factor 1:', d['enact_ms_per_step'], d['value'])"
python tools/fuzz_others.py 30 808 2>&1 | tail -1
