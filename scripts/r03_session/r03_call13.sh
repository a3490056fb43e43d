set -x
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_scanline.py tests/test_gpu_pgm.py -q -m gpu -x > gpurun_out/r03/pytest13.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r03/pytest13.log
timeout -k 10 300 python scripts/scanline_probe.py > gpurun_out/r03/scanline_probe4.txt 2>&1; cat gpurun_out/r03/scanline_probe4.txt
timeout -k 10 300 python bench.py --workload basic --no-cpu-baseline > gpurun_out/r03/bench_basic2.json 2> gpurun_out/r03/bench_basic2.err; head -c 600 gpurun_out/r03/bench_basic2.json; echo
timeout -k 10 300 python bench.py --strong-in-process --no-cpu-baseline --no-ar-workloads --no-dominant > gpurun_out/r03/bench_strong_inproc.json 2> gpurun_out/r03/bench_strong_inproc.err; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_strong_inproc.json') if l.startswith('{')][-1])
print('value', d['value'], 'strong', d.get('strong'), 'proxy', d.get('strong_per_gpu_proxy',{}).get('value'))"
bash scripts/kodak_workers.sh gpurun_out/r03 > gpurun_out/r03/kodak_workers.log 2>&1; cat gpurun_out/r03/kodak_workers/summary.txt
