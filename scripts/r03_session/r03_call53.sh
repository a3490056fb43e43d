for cfg in "3 2" "3 3" "4 2" "4 3" "3 2" "4 2"; do set -- $cfg
  timeout -k 10 200 python bench.py --workers $1 --token-lanes $2 --steps 24 --no-cpu-baseline --no-extra-legs --no-dominant --no-ar-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('workers $1 token lanes $2 batch 256:', round(d['value'],1), round(d['ms_per_step'],2))"
done
