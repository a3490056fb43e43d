mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_codec.py tests/test_gpu_codec_graph.py -q -m gpu -x > gpurun_out/r03/pytest40.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r03/pytest40.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --no-ar-workloads --no-cpu-baseline > gpurun_out/r03/bench_gcq.json 2> gpurun_out/r03/bench_gcq.err; python -c "
import json; d=json.loads(open('gpurun_out/r03/bench_gcq.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('end_to_end',{}).get('frac'), d.get('strong_per_gpu_proxy',{}).get('value'))"
