set -x
mkdir -p gpurun_out/r03
timeout -k 10 300 python scripts/mconv_probe.py default: nostage:BASIC_MCONV_DEBUG=1 > gpurun_out/r03/mconv_probe4.txt 2>&1
PROBE_PERM=1 timeout -k 10 300 python scripts/mconv_probe.py perm: perm_nostage:BASIC_MCONV_DEBUG=1 perm_nostore:BASIC_MCONV_DEBUG=4 >> gpurun_out/r03/mconv_probe4.txt 2>&1; cat gpurun_out/r03/mconv_probe4.txt
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_pgm.py tests/test_gpu_ar_codecs.py tests/test_gpu_scanline.py tests/test_gpu_codec_graph.py tests/test_gpu_reference_kats.py -q -m gpu -x > gpurun_out/r03/pytest6.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03/pytest6.log
rm -f gpurun_out/r03/lanes_b32b.txt
for cfg in "5 3" "6 3" "6 4" "5 0" "6 0"; do set -- $cfg; echo "== batch 32 workers $1 token-lanes $2" >> gpurun_out/r03/lanes_b32b.txt; timeout -k 10 200 python bench.py --batch 32 --workers $1 --token-lanes $2 --steps 96 --no-cpu-baseline --no-extra-legs --no-dominant 2>>gpurun_out/r03/lanes.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['end_to_end']['frac'],3), 'call', round(d['config'].get('call_latency_ms',0),1))" >> gpurun_out/r03/lanes_b32b.txt; done; cat gpurun_out/r03/lanes_b32b.txt
timeout -k 10 300 python bench.py --workload checkerboard --no-cpu-baseline > gpurun_out/r03/bench_cb3.json 2> gpurun_out/r03/bench_cb3.err; head -c 1100 gpurun_out/r03/bench_cb3.json
