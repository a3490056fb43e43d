mkdir -p gpurun_out/r03
timeout -k 10 600 env BASIC_CONV_WAVES25=4 python -m pytest tests/test_gpu_conv.py -q -m gpu -x > gpurun_out/r03/pytest51.log 2>&1; rc=$?; echo "pytest (4-wave packing) rc $rc"; tail -2 gpurun_out/r03/pytest51.log
[ $rc -eq 0 ] || exit 1
for w in 0 4; do
  if [ $w = 4 ]; then export BASIC_CONV_WAVES25=4; else unset BASIC_CONV_WAVES25; fi
  for i in 1 2; do
  timeout -k 10 200 python bench.py --batch 32 --workers 6 --token-lanes 4 --steps 96 --warmup 6 --no-cpu-baseline --no-extra-legs --no-dominant 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('waves25=$w batch 32:', round(d['value'],1), round(d['ms_per_step'],2))"
  done
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs --no-dominant --no-ar-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('waves25=$w batch 256:', round(d['value'],1), round(d['ms_per_step'],2))"
done
