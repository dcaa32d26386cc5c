for cfg in "128 3 4" "128 4 4" "96 4 4" "160 3 4" "192 2 4" "256 2 4" "128 3 8" "128 3 2" "64 6 4"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --no-latency --no-roofline --slots $1 --engines $2 --chunk $3 --pairs-per-step 4096 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('slots $1 engines $2 chunk $3 ->', round(d['value'],1), round(d['ms_per_step'],1))"
done
