for cfg in "32 6 4" "32 6 5" "32 6 6" "256 6 4" "256 6 5" "256 6 6"; do set -- $cfg
  st=24; [ $1 = 32 ] && st=96
  timeout -k 10 250 python bench.py --batch $1 --workers $2 --token-lanes $3 --steps $st --warmup 6 --no-cpu-baseline --no-extra-legs --no-dominant --no-ar-workloads 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('batch $1 workers $2 lanes $3:', round(d['value'],1), round(d['ms_per_step'],2))
    elif 'rror' in l: print(l.strip()[:200])"
done
