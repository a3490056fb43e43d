for rw in 8 4 16 8; do
  timeout -k 10 250 python bench.py --rans-waves $rw --steps 24 --no-cpu-baseline --no-extra-legs --no-dominant --no-ar-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('workers 6 lanes 4 rans-waves $rw:', round(d['value'],1), round(d['ms_per_step'],2))"
done
for w in 7 8; do
  timeout -k 10 250 python bench.py --workers $w --token-lanes 4 --steps 24 --no-cpu-baseline --no-extra-legs --no-dominant --no-ar-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('workers $w lanes 4:', round(d['value'],1), round(d['ms_per_step'],2))"
done
