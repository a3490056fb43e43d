mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_scanline.py tests/test_gpu_ar_codecs.py -q -m gpu -x > gpurun_out/r03/pytest50.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r03/pytest50.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do timeout -k 10 300 python bench.py --workload basic --no-cpu-baseline > gpurun_out/r03/bench_basic_u32.json 2> gpurun_out/r03/bench_basic_u32.err; python -c "
import json; d=json.loads(open('gpurun_out/r03/bench_basic_u32.json').read().strip().splitlines()[-1]); print('basic:', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'avg launch ms', round(d['roofline']['avg_launch_ms']*1000,1), 'us')"; done
