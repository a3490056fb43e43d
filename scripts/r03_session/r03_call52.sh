for l in 1 2 1 2; do
  timeout -k 10 200 python bench.py --token-lanes $l --steps 24 --no-cpu-baseline --no-extra-legs --no-dominant --no-ar-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('token lanes $l batch 256:', round(d['value'],1), round(d['ms_per_step'],2))"
done
