mkdir -p gpurun_out/r03
rm -f gpurun_out/r03/lanes_b32c.txt
for cfg in "6 4 8 8" "6 4 4 8" "6 4 16 8" "5 4 8 8" "6 4 8 6" "6 4 8 7" "6 3 4 8"; do set -- $cfg; echo "== batch 32 workers $1 token-lanes $2 rans-waves $3 hw-queues $4" >> gpurun_out/r03/lanes_b32c.txt; GPU_MAX_HW_QUEUES=$4 timeout -k 10 200 python bench.py --batch 32 --workers $1 --token-lanes $2 --rans-waves $3 --steps 96 --warmup 6 --no-cpu-baseline --no-extra-legs --no-dominant 2>>gpurun_out/r03/lanes.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['end_to_end']['frac'],3), 'call', round(d['config'].get('call_latency_ms',0),1))" >> gpurun_out/r03/lanes_b32c.txt; done; cat gpurun_out/r03/lanes_b32c.txt
timeout -k 10 400 python bench.py --gpus 2 --dist-backend gloo --same-device --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r03/rehearsal_gpus2.json 2> gpurun_out/r03/rehearsal_gpus2.err; tail -c 900 gpurun_out/r03/rehearsal_gpus2.json; tail -3 gpurun_out/r03/rehearsal_gpus2.err
