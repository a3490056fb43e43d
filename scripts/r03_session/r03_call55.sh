for cfg in "3 1" "4 3" "6 4" "4 3" "6 4" "3 1"; do set -- $cfg
  timeout -k 10 250 python bench.py --workers $1 --token-lanes $2 --no-cpu-baseline --no-extra-legs --no-dominant --no-ar-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('workers $1 token lanes $2 batch 256 default steps', d['steps'], ':', round(d['value'],1), round(d['ms_per_step'],2))"
done
for cfg in "4 3" "6 4"; do set -- $cfg
  timeout -k 10 250 python bench.py --workers $1 --token-lanes $2 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs --no-dominant --no-ar-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('workers $1 token lanes $2 batch 256 steps 5:', round(d['value'],1), round(d['ms_per_step'],2))"
done
