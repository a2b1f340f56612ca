for cfg in "9000000 30000000" "9000000 40000000" "9000000 48000000" "6000000 34000000" "12000000 34000000" "9000000 34000000 --lite-factor 6" "9000000 34000000 --lite-factor 24"; do
  set -- $cfg
  mn=$1; mx=$2; shift 2
  timeout -k 10 300 python bench.py --no-cpu-baseline --skip-topdown-leg --head-pass-min $mn --head-pass-max $mx "$@" > gpurun_out/hp.log 2>&1 || { echo "$cfg FAILED"; continue; }
  tail -1 gpurun_out/hp.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['enact_ms_per_step'], d['roofline']['frac'])"
done
