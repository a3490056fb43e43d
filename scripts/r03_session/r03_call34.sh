mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_scanline.py tests/test_gpu_ar_codecs.py tests/test_gpu_harness_workers.py -q -m gpu -x > gpurun_out/r03/pytest34.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r03/pytest34.log
[ $rc -eq 0 ] || exit 1
KODAK_CFGS="basic:0 basic:3 basic:3 basic:6" bash scripts/kodak_workers.sh gpurun_out/r03 > gpurun_out/r03/kodak_workers8.log 2>&1; cat gpurun_out/r03/kodak_workers/summary.txt
timeout -k 10 300 python bench.py --workload basic --no-cpu-baseline > gpurun_out/r03/bench_basic.json 2> gpurun_out/r03/bench_basic.err; python -c "
import json; d=json.loads(open('gpurun_out/r03/bench_basic.json').read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], d['config'])"
