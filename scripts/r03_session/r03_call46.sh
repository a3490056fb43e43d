mkdir -p gpurun_out/r03
rm -f gpurun_out/r03/lanes_b32d.txt
run() { # workers lanes waves noside
  echo "== batch 32 workers $1 token-lanes $2 rans-waves $3 no-side-stream $4" >> gpurun_out/r03/lanes_b32d.txt
  if [ "$4" = 1 ]; then export BASIC_HP_NO_SIDE_STREAM=1; else unset BASIC_HP_NO_SIDE_STREAM; fi
  GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python bench.py --batch 32 --workers $1 --token-lanes $2 --rans-waves $3 --steps 96 --warmup 6 --no-cpu-baseline --no-extra-legs --no-dominant 2>>gpurun_out/r03/lanes.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['end_to_end']['frac'],3), 'call', round(d['config'].get('call_latency_ms',0),1))" >> gpurun_out/r03/lanes_b32d.txt
}
run 6 4 8 0
run 6 4 8 1
run 7 4 8 1
run 8 4 8 1
run 8 4 4 1
run 7 3 8 1
cat gpurun_out/r03/lanes_b32d.txt
