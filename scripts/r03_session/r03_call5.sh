set -x
mkdir -p gpurun_out/r03
timeout -k 10 300 python scripts/mconv_probe.py default: nostage:BASIC_MCONV_DEBUG=1 nostore:BASIC_MCONV_DEBUG=4 > gpurun_out/r03/mconv_probe3.txt 2>&1; cat gpurun_out/r03/mconv_probe3.txt
timeout -k 10 300 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "masked" > gpurun_out/r03/pytest5.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03/pytest5.log
for cfg in "3 1" "3 2" "4 1" "4 2" "5 2" "6 2"; do set -- $cfg; echo "== batch 32 workers $1 token-lanes $2" >> gpurun_out/r03/lanes_b32.txt; timeout -k 10 200 python bench.py --batch 32 --workers $1 --token-lanes $2 --steps 96 --no-cpu-baseline --no-extra-legs --no-dominant 2>>gpurun_out/r03/lanes.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['end_to_end']['frac'],3), 'call', round(d['config'].get('call_latency_ms',0),1))" >> gpurun_out/r03/lanes_b32.txt; done; cat gpurun_out/r03/lanes_b32.txt
timeout -k 10 200 python bench.py --token-lanes 2 --steps 24 --no-cpu-baseline --no-extra-legs --no-dominant 2>>gpurun_out/r03/lanes.err | cut -c1-200
